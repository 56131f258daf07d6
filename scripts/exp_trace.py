"""Where do the workgroups of the C2 launch run and when do they finish?  Needs the trace build
(make EXTRA=-DPRALINE_TRACE; PRALINE_LIB=scripts/micro/trace.bin)."""
import sys, os, ctypes, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from praline_amd import native as nat
from praline_amd.matrices import blosum62_matrix
from bench import synth_lengths, synth_profile
nat.init(0)
L = nat.lib()
N = int(os.environ.get("N", "256"))
rng = np.random.default_rng(2)
lens = synth_lengths(rng, N, 400)
profs = [synth_profile(rng, int(x)) for x in lens]
pairs = np.array([(i, j) for i in range(N) for j in range(i + 1, N)], dtype=np.int32)
ar = nat.Arena(profs, blosum62_matrix()); pl = nat.Plan(ar, pairs)
buf = torch.zeros(4096 * 4 * 6, dtype=torch.int64, device="cuda")
L.praline_trace_set.argtypes = [ctypes.c_void_p]
assert L.praline_trace_set(ctypes.c_void_p(buf.data_ptr())) == 0
for _ in range(3): pl.run("global", -11, -1)
nat.synchronize(); torch.cuda.synchronize()
buf.zero_(); torch.cuda.synchronize()
pl.run("global", -11, -1); nat.synchronize(); torch.cuda.synchronize()
print("kernel_ms", pl.kernel_ms())
r = buf.cpu().numpy().reshape(-1, 6)
r = r[r[:, 5] != 0]
blk, wv = r[:, 0], r[:, 1] & 0xff
share = (r[:, 1] >> 8) & 0xff
hw = r[:, 2]
xcc = r[:, 3] & 0xf
simd = (hw >> 4) & 0x3; cu = (hw >> 8) & 0xf; sh = (hw >> 12) & 0x1; se = (hw >> 13) & 0x7
# s_memtime counts core clocks here and is not synchronised between XCDs: normalise per XCD
start = np.zeros(len(r)); end = np.zeros(len(r))
for x in np.unique(xcc):
    m = xcc == x
    t0 = r[m, 4].min()
    start[m] = (r[m, 4] - t0) / 1e6   # Mcycles
    end[m] = (r[m, 5] - t0) / 1e6
print("waves recorded", len(r), " span %.2f Mcycles -> %.2f GHz" % (end.max(), end.max() / pl.kernel_ms() * 1e-3 * 1e3))
cu_key = xcc * 1000 + se * 100 + sh * 50 + cu
print("distinct (xcc,se,sh,cu):", len(np.unique(cu_key)), " distinct xcc:", np.unique(xcc))
# which blocks share a CU
from collections import defaultdict
d = defaultdict(list)
for b, k in zip(blk, cu_key): d[k].append(int(b))
samples = list(d.items())[:6]
for k, v in samples: print("cu", k, "blocks", sorted(set(v)))
# per-SIMD busy: waves per (cu, simd) and their end times
sk = cu_key * 4 + simd
ends = defaultdict(list)
for k, s_, e in zip(sk, start, end): ends[k].append((s_, e))
last = np.array([max(e for _, e in v) for v in ends.values()])
nw = np.array([len(v) for v in ends.values()])
print("SIMDs used", len(ends), " waves/SIMD min/mean/max", nw.min(), nw.mean(), nw.max())
print("SIMD finish (Mcycles): min %.2f  p10 %.2f  median %.2f  p90 %.2f  max %.2f" % (last.min(), np.percentile(last, 10), np.median(last), np.percentile(last, 90), last.max()))
print("late starters (start > 0.05 Mcycles):", int((start > 0.05).sum()), " of", len(start)); busy = np.array([sum(e - s_ for s_, e in v) for v in ends.values()]); print("per-SIMD sum of wave durations (Mcycles): min %.2f median %.2f max %.2f ; mean finish %.2f" % (busy.min(), np.median(busy), busy.max(), last.mean()))
dur = end - start
for sh_ in (1, 2, 4):
    m = share == sh_
    if m.any(): print("share %d: %d waves, duration Mcycles min/median/max %.2f %.2f %.2f" % (sh_, m.sum(), dur[m].min(), np.median(dur[m]), dur[m].max()))
