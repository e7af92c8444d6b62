"""usage: python tools/capture_probe.py cold|warm  -- on a GPU box: an NL and a TL launch recorded into a HIP graph while the stream is
capturing (torch.cuda.graph), replayed and compared with the eager launches.  cold: the captured launches are the first of the
process (the level table of this CETA is created by the capturing launcher itself); warm: one eager NL launch first."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import dwarf_p_cloudsc2_tl_ad_amd as c2  # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else "cold"
tab = c2.synthetic_table()
prm = c2.default_params(c2.ceta_from_table(tab), lregcl=True)
ds = c2.DeviceState.from_table(tab, 128, 140000)  # one round of NL waves with unequal SIMD loads; 2 rounds + 70 for TL: both heuristics apply
ds.satur(prm)
dx = ds.increments(zero_supsat=True)
dy = c2.FlatFields("out", ds.nb, ds.nlev, ds.nproma, ds.device)
if mode == "warm":
    ds.nl(prm)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
with torch.cuda.graph(g, stream=s):
    cs = torch.cuda.current_stream()
    ds.nl(prm, stream=cs)
    ds.satur(prm, stream=cs)
    ds.tl(prm, dx, dy, stream=cs)
outs = (ds.B_LOC, ds.PA, ds.PCOVPTOT, ds.PFPLSL, ds.PFPLSN, ds.PFHPSL, ds.PFHPSN)


def poison():
    for t in outs:
        t.fill_(-7.0)
    for k in dy.t:
        dy.t[k].fill_(-7.0)


poison()
g.replay()
torch.cuda.synchronize()
a = [t.clone() for t in outs] + [dy.t[k].clone() for k in sorted(dy.t)]
assert not any(bool((t == -7.0).all()) for t in a[:7]), "the replay wrote nothing"
poison()
ds.nl(prm)
ds.satur(prm)
ds.tl(prm, dx, dy)
torch.cuda.synchronize()
b = list(outs) + [dy.t[k] for k in sorted(dy.t)]
assert all(torch.equal(x, y) for x, y in zip(a, b)), "replayed graph and eager launches differ"
print("CAPTURE OK", mode, len(a))
