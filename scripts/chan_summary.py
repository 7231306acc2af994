#!/usr/bin/env python3
"""Per-channel L2 -> fabric request counters of every kr_scan launch of a `rocprofv3 --pmc TCC_EA0_* --output-format json` run
(scripts/r3_session_e.sh): duration next to the spread of TCC_EA0_RDREQ over the 16 channels x 8 XCDs and the mean read
latency (TCC_EA0_RDREQ_LEVEL / TCC_EA0_RDREQ) -- is a slow launch a channel imbalance?"""
import glob, json, sys, collections
root = sys.argv[1]
files = glob.glob(root + "/**/*results.json", recursive=True) + glob.glob(root + "/**/*.json", recursive=True)
f = sorted(set(files), key=lambda p: -len(p))[0] if files else None
print("file:", f)
d = json.load(open(f))
tool = d["rocprofiler-sdk-tool"][0]
print("keys:", list(tool.keys()))
br = tool["buffer_records"]
print("buffer_records:", {k: len(v) for k, v in br.items() if isinstance(v, list)})
cb = tool.get("callback_records", {})
print("callback_records:", {k: len(v) for k, v in cb.items() if isinstance(v, list)})
# kernel names
ksym = {}
for k in tool.get("kernel_symbols", []):
    ksym[k.get("kernel_id")] = k.get("formatted_kernel_name") or k.get("kernel_name")
counters = {c["id"]["handle"] if isinstance(c.get("id"), dict) else c.get("id"): c for c in tool.get("counters", [])}
print("ncounters meta:", len(counters))
cc = cb.get("counter_collection") or br.get("counter_collection") or []
print("counter_collection records:", len(cc))
if cc:
    print("sample record keys:", list(cc[0].keys()))
    print(json.dumps(cc[0])[:1500])
disp_t = {}
for kd in br.get("kernel_dispatch", []):
    di = kd["dispatch_info"]
    disp_t[di["dispatch_id"]] = (kd["end_timestamp"] - kd["start_timestamp"], ksym.get(di["kernel_id"], "?"))
rows = []
for rec in cc:
    di = rec["dispatch_data"]["dispatch_info"]
    name = ksym.get(di["kernel_id"], "?")
    if "kr_scan" not in name:
        continue
    per = collections.defaultdict(list)
    for r in rec["records"]:
        cid = r["counter_id"]["handle"] if isinstance(r["counter_id"], dict) else r["counter_id"]
        meta = counters.get(cid, {})
        per[meta.get("name", str(cid))].append(r["value"])
    dur = disp_t.get(di["dispatch_id"], (0, ""))[0]
    rows.append((di["dispatch_id"], dur, per))
import statistics as st
for did, dur, per in rows:
    out = [f"dispatch {did} dur_ms {dur/1e6:8.2f}"]
    for k, v in sorted(per.items()):
        if len(v) > 1:
            out.append(f"{k}: n={len(v)} sum={sum(v):.4g} min={min(v):.4g} max={max(v):.4g} cv={st.pstdev(v)/max(1e-9,st.mean(v)):.3f}")
        else:
            out.append(f"{k}: {v[0]:.4g}")
    rd, lv = per.get("TCC_EA0_RDREQ"), per.get("TCC_EA0_RDREQ_LEVEL")
    if rd and lv and len(rd) == len(lv):
        lat = [l / max(1.0, r) for r, l in zip(rd, lv)]
        out.append(f"latency(cycles): mean={sum(lv)/max(1.0,sum(rd)):.1f} min={min(lat):.1f} max={max(lat):.1f}")
    print(" | ".join(out))
