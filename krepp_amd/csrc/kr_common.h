// kr_common.h — internal declarations shared by the host and device translation units.
#ifndef KR_COMMON_H
#define KR_COMMON_H

#include "krepp_amd.h"

#include <string>
#include <vector>

namespace kr {

// per-thread error message behind kr_last_error()
int fail(int code, const std::string& msg);
void clear_error();

struct TreeNode {
  std::string label;   // as written in the Newick / reflist ("" if unlabelled)
  double blen;         // NaN if absent
  uint32_t parent;     // se of the parent, 0 for the root
  uint8_t kind;        // 1 leaf, 2 internal
};

// Tree with the reference's numbering: se = 1-based post-order index assigned while
// parsing (src/phytree.cpp:168-170,211-213); nodes[0] is the null node.
struct HostTree {
  std::vector<TreeNode> nodes;
  uint32_t nnodes() const { return (uint32_t)nodes.size() - 1; }
  std::string name(uint32_t se) const; // Node::get_name, src/phytree.hpp:134-145
};

bool parse_newick(const std::string& text, HostTree& out, std::string& err);
void balanced_tree(const std::vector<std::string>& names, HostTree& out);

// MurmurHash3_x86_32 based 64-bit name hash (src/record.hpp:26-36)
uint64_t leaf_name_hash(const std::string& name);
uint64_t rehash64(uint64_t sh); // Subset::rehash, src/record.hpp:37-47

} // namespace kr

#endif
