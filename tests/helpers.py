"""Test helpers: an independent pure-Python closed form of the integer front end
 and a hand-written index writer for crafted scenarios."""
import os
import struct

import numpy as np

CODE = {c: i for i, c in enumerate("ACGT")}
CODE.update({c.lower(): i for c, i in list(CODE.items())})


def kmer_codes(kmer: str):
    """c[p] = code of base s[k-1-p] (p = 0 is the LAST base), or None if any base is not ACGT."""
    if any(ch not in CODE for ch in kmer):
        return None
    return [CODE[ch] for ch in reversed(kmer)]


def revcomp(s: str):
    comp = {"A": "T", "C": "G", "G": "C", "T": "A", "a": "t", "c": "g", "g": "c", "t": "a"}
    return "".join(comp.get(ch, "N") for ch in reversed(s))


def closed_form(kmer: str, ppos, npos):
    """(enc_bp, enc_lr, rix, enc32) from the position lists alone."""
    c = kmer_codes(kmer)
    if c is None:
        return None
    enc_bp = sum(cp << (2 * p) for p, cp in enumerate(c))
    enc_lr = sum(((cp >> 1) << (p + 32)) | ((cp & 1) << p) for p, cp in enumerate(c))
    P = sorted(int(x) for x in ppos)
    N = sorted(int(x) for x in npos)
    rix = sum(c[P[j]] << (2 * j) for j in range(len(P)))
    enc32 = sum(((c[N[j]] & 1) << j) | ((c[N[j]] >> 1) << (16 + j)) for j in range(len(N)))
    return enc_bp, enc_lr, rix, enc32


def hd32(a, b):
    z = a ^ b
    return bin((z | (z >> 16)) & 0xFFFF).count("1")


def front_end_py(seq: str, k, ppos, npos):
    """Every (kpos, strand) with a fully valid window, as the reference enumerates them."""
    out = []
    for i in range(0, len(seq) - k + 1):
        km = seq[i:i + k]
        f = closed_form(km, ppos, npos)
        if f is None:
            continue
        r = closed_form(revcomp(km), ppos, npos)
        out.append((i, 0) + f)
        out.append((i, 1) + r)
    return out


def write_index(path, k, h, m, r, frac, ppos, rows, pse, rho, nwk=None, reflist=None, w=None, nrows=None):
    """rows: {row: [(enc32, se), ...]} (sorted by enc32 here).  pse: list of (a, b) with pse[0]=(0,0)."""
    os.makedirs(path, exist_ok=True)
    sfx = f"-m{m}r{r}-" + ("frac" if frac else "no_frac")
    ppos = sorted(ppos, reverse=True)
    npos = [p for p in range(k) if p not in ppos]
    if nrows is None:
        hs = 4 ** h
        nrows = (hs // m) * (r + 1) if frac else hs // m
    inc = np.zeros(nrows, np.uint64)
    cm = []
    for row in range(nrows):
        ent = sorted(rows.get(row, []))
        cm.extend(ent)
        inc[row] = len(cm)
    with open(os.path.join(path, "cmer" + sfx), "wb") as f:
        f.write(struct.pack("<Q", len(cm)))
        f.write(np.array(cm, dtype=np.uint32).reshape(-1, 2).tobytes() if cm else b"")
    with open(os.path.join(path, "inc" + sfx), "wb") as f:
        f.write(struct.pack("<I", nrows))
        f.write(inc.tobytes())
    with open(os.path.join(path, "crecord" + sfx), "wb") as f:
        f.write(struct.pack("<II", len(rho), len(pse)))
        f.write(np.array(pse, dtype=np.uint32).reshape(-1, 2).tobytes())
        f.write(np.array(rho, dtype=np.float64).tobytes())
    with open(os.path.join(path, "metadata" + sfx), "wb") as f:
        f.write(struct.pack("<BBBIIBI", k, w or k + 6, h, m, r, 1 if frac else 0, nrows))
        f.write(bytes(ppos) + bytes(npos))
    if nwk is not None:
        open(os.path.join(path, "tree" + sfx), "w").write(nwk)
    if reflist is not None:
        open(os.path.join(path, "reflist" + sfx), "w").write("".join(n + "\n" for n in reflist))
    return sfx


def row_of(rix, m, r, frac):
    """Index::bucket_indices (src/index.cpp:160-168) / RSeq::extract_mers row (src/rqseq.cpp:125-128)."""
    res, q = rix % m, rix // m
    if frac:
        return q * (r + 1) + res if res <= r else None
    return q if res == r else None
