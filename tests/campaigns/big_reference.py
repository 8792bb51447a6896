import sys, time, os
sys.path.insert(0, os.getcwd())
import numpy as np
import sigfish_amd as S
from sigfish_amd import synth, api
MB = int(sys.argv[1]) if len(sys.argv) > 1 else 32
t=time.time()
lv = synth.kmer_levels(6, 1)
seq = synth.random_sequence(MB * 1_000_000, 11)
ref = api.RefModel.from_records([("big", seq)], lv, 6, 0, 250)
print("ref built", time.time()-t, ref.ref_lengths, flush=True)
n = 64
q, q_off, truth = synth.make_reads(ref, n, qlen=250, seed=5, short_frac=0.0)
print("reads made", time.time()-t, flush=True)
with S.Aligner(ref, 0) as al:
    print("ctx", time.time()-t, flush=True)
    t1=time.time(); a = al.align_db(q, q_off); print("align 1", time.time()-t1, flush=True)
    t1=time.time(); b = al.align_db(q, q_off); print("align 2", time.time()-t1, al.profile(), flush=True)
assert a.tobytes()==b.tobytes()
rl = int(ref.ref_lengths[0])
exp_st = np.where(truth["strand"] == 0, truth["start"], rl - (truth["start"] + truth["span"]))
print("near truth:", (np.abs(a["pos_st"] - exp_st) <= 30).mean(), "valid", a["valid"].mean())
# cut-out property
ok = 0; tried = 0
W = 250 * 2000
for i in range(n):
    st = int(a["pos_st"][i])
    fwd = a["strand"][i] == ord("+")
    # window start in the coordinates of the strand's own array, multiple of 250, best locus in the middle
    own = st if fwd else rl - int(a["pos_end"][i])  # column of the hit in its strand's own array (approximately: the window is wide)
    lo = max(0, (own - W // 2) // 250 * 250); hi = min(rl, lo + W)
    f = ref.forward[0]; r = ref.reverse[0]
    if fwd:
        sub_f = f[lo:hi]; sub_r = r[rl - hi: rl - lo]
    else:
        sub_r = r[lo:hi]; sub_f = f[rl - hi: rl - lo]
    sub = api.RefModel(["sub"], [hi - lo + 5], [hi - lo], [0], [np.ascontiguousarray(sub_f)], [np.ascontiguousarray(sub_r)])
    with S.Aligner(sub, 0) as al2:
        c = al2.align_db(q[q_off[i]:q_off[i+1]], np.array([0, 250], np.int64))
    tried += 1
    shift = lo if fwd else (rl - hi)  # positions are reported on the forward strand
    same = (c["score"][0].tobytes() == a["score"][i].tobytes() and c["strand"][0] == a["strand"][i]
            and int(c["pos_st"][0]) + shift == st and int(c["pos_end"][0]) + shift == int(a["pos_end"][i]))
    ok += same
    if not same: print("diff", i, a[i], c[0], lo)
    if tried >= 12: break
print("cut-out equal:", ok, "of", tried)
