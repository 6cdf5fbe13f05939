"""Dev aid: run N seeded blocks through HIP engine and oracle, print mismatching fields."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from oracle import thrifty_np as onp
from thrifty_amd import _native as F, synth

n, h = 16384, 4096
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
tpl = synth.gold_template(10, 2)
win = onp.unique_window(n, h, len(tpl))
rng = np.random.default_rng(4242)
blocks, _ = synth.synth_blocks(rng, nb, n, tpl, win)
eng = F.Engine(n, h, tpl, (0, 15, 0), (7, 110), (0, 15, 0), max_batch=1024)
recs = [eng.detect(blocks)[:, 0] for _ in range(3)]
for k in (1, 2):
    same = all(np.array_equal(recs[0][f], recs[k][f]) for f in recs[0].dtype.names)
    print("run 0 vs run %d identical: %s" % (k, same))
    if not same:
        for f in recs[0].dtype.names:
            d = np.flatnonzero(recs[0][f] != recs[k][f])
            if len(d):
                print("  field", f, "differs at", d[:10], recs[0][f][d[:5]], recs[k][f][d[:5]])
orc = onp.OracleDetector(n, h, tpl, (0, 15, 0), (7, 110), (0, 15, 0))
bad = 0
for i in range(nb):
    (res,) = orc.detect_u8(i, blocks[i])
    r = recs[0][i]
    msgs = []
    if r["carrier_bin"] != res.carrier.bin: msgs.append("cbin %d vs %d" % (r["carrier_bin"], res.carrier.bin))
    if res.carrier.detected:
        if r["corr_sample"] != res.corr.sample: msgs.append("sample %d vs %d" % (r["corr_sample"], res.corr.sample))
        if abs(r["corr_energy"] - res.corr.energy) > 1e-4 * res.corr.energy: msgs.append("energy %r vs %r" % (r["corr_energy"], res.corr.energy))
        if abs(r["corr_offset"] - res.corr.offset) > 1e-4: msgs.append("offset %r vs %r" % (r["corr_offset"], res.corr.offset))
        if abs(r["carrier_offset"] - res.carrier.offset) > 1e-4: msgs.append("coffset %r vs %r" % (r["carrier_offset"], res.carrier.offset))
    if msgs:
        bad += 1
        print(i, "; ".join(msgs))
print("mismatching blocks:", bad, "of", nb)
