"""Quick GPU parity probe (development aid; the real checks live in tests/)."""
import sys, time, os, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'oracle'))
from krepp_amd import capi, synth
import pyoracle as po

nreads = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
k, w, h = (int(x) for x in (sys.argv[2:5] if len(sys.argv) > 4 else (21, 27, 7)))
glen = int(sys.argv[5]) if len(sys.argv) > 5 else 20000
nwk = open(os.path.join(ROOT, 'tests/golden/tree_toy.nwk')).read()
g = synth.evolve_genomes(nwk, glen, seed=7)
tsv = synth.write_genomes(g, '/tmp/toy_g')
capi.build_index(tsv, '/tmp/toy_idx', nwk=os.path.join(ROOT, 'tests/golden/tree_toy.nwk'), k=k, w=w, h=h, m=4, r=1, frac=True, num_threads=8)
hx = capi.HostIndex('/tmp/toy_idx')
ox = po.Index('/tmp/toy_idx')
b, o, names = synth.sample_reads(g, nreads, seed=1)
t = time.time(); ref = ox.dist(b, o, names, po.params(collect=7)); t_or = time.time() - t
print('oracle s', t_or, ref['counters'])
dx = hx.upload(0)
print('device bytes', dx.device_bytes)
# front end
stride = 150 - k + 1
rix, enc, valid, pas = dx.front_end(b, o, stride)
bad = 0
for r in range(min(nreads, 300)):
    fe = ox.front_end(b[int(o[r]):int(o[r+1])].tobytes())
    for i in range(len(fe['kpos'])):
        kp, s = int(fe['kpos'][i]), int(fe['strand'][i])
        if not valid[r, kp, s] or rix[r, kp, s] != fe['rix'][i] or enc[r, kp, s] != fe['enc32'][i] or pas[r, kp, s] != fe['pas'][i]:
            bad += 1
            if bad < 5: print('FE mismatch', r, kp, s, valid[r,kp,s], hex(rix[r,kp,s]), hex(fe['rix'][i]), hex(enc[r,kp,s]), hex(fe['enc32'][i]))
    nval = int(valid[r, :, 0].sum())
    assert nval == len(fe['kpos']) // 2, (r, nval, len(fe['kpos']))
print('front-end mismatches', bad)
st = dx.stream(max_reads=nreads, max_bases=len(b))
st.submit(b, o, capi.KR_TAP_ACCS | capi.KR_TAP_HITS)
res = st.collect()
tm = st.timing()
print('gpu ms scan', tm.ms_scan, 'acc', tm.ms_acc, 'llh', tm.ms_llh, 'ovf reads', tm.overflow_reads, 'nrecs', res.nrecs, 'nrows', res.nrows)
# hits
gh = st.hits()
a = sorted(zip(gh['read'].tolist(), gh['strand'].tolist(), gh['kpos'].tolist(), gh['cmer_index'].tolist(), gh['hd'].tolist()))
rh = ref['hits']
bset = sorted(zip(rh['read'].tolist(), rh['strand'].tolist(), rh['kpos'].tolist(), rh['cmer_index'].tolist(), rh['hd'].tolist()))
print('hits gpu', len(a), 'oracle', len(bset), 'equal', a == bset)
# accs (passed only)
acc = ref['accs']; accp = acc[acc['passed'] == 1]
ok = sorted(zip(accp['read'].tolist(), ((accp['se'] << 1) | accp['strand']).tolist(), [tuple(x[:5]) for x in accp['hist'].tolist()]))
gk = sorted(zip(res.rec_read.tolist(), res.rec_key.tolist(), [tuple(x) for x in res.rec_hist.tolist()]))
print('accs gpu', len(gk), 'oracle', len(ok), 'equal', gk == ok)
# onmers / filt
ri = ref['reads']
print('onmers equal', bool((ri['onmers'] == res.read_onmers).all()), 'filt equal', bool((ri['hdist_filt'] == st.readtaps(nreads)).all()))
# d
if gk == ok:
    od = {(int(r_), int((s_ << 1) | t_)): (d_, v_) for r_, s_, t_, d_, v_ in zip(accp['read'], accp['se'], accp['strand'], accp['d_llh'], accp['v_llh'])}
    rel = [abs(res.rec_d[i] - od[(int(res.rec_read[i]), int(res.rec_key[i]))][0]) / od[(int(res.rec_read[i]), int(res.rec_key[i]))][0] for i in range(res.nrecs)]
    print('max rel d err', max(rel) if rel else 0, 'n>1e-6', sum(x > 1e-6 for x in rel), 'n>0', sum(x > 0 for x in rel))
# rows
grow = res.rows()
orow = sorted((int(r_), int(s_), float(d_)) for r_, s_, d_ in zip(ref['rows']['read'], ref['rows']['se'], ref['rows']['d_llh']) if s_ != 0)
print('rows gpu', len(grow), 'oracle', len(orow), 'same keys', [x[:2] for x in grow] == [x[:2] for x in orow])
txt = st.format_dist(hx, names)
print('text equal', sorted(txt.splitlines()) == sorted(ref['text'].splitlines()))
