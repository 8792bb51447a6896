import time, sys, os
sys.path.insert(0, os.getcwd())
t0=time.time()
import numpy as np
import sigfish_amd as S
from sigfish_amd import synth
t1=time.time()
ref, flag, q, q_off, meta = synth.workload("ncov_r9_dna_q250", n_reads=4096, seed=7)
t2=time.time()
al=S.Aligner(ref, flag, device=0)
t3=time.time()
al2=S.Aligner(ref, flag, device=0)
t4=time.time()
r=al.align_db(q,q_off)
t5=time.time()
r=al.align_db(q,q_off)
t6=time.time()
r=al2.align_db(q,q_off)
t7=time.time()
print(f"import {t1-t0:.3f}  first context {t3-t2:.3f}  second context {t4-t3:.3f}  first batch {t5-t4:.3f}  second batch {t6-t5:.3f}  first batch on ctx2 {t7-t6:.3f}")
