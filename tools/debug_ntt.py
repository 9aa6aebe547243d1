import sys, random, numpy as np
sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import oracle_lib as O
from lambdaworks_cairo_prover_amd import api
ctx = api.Context()
for k in [18,19,20,21,22]:
    rng = random.Random(k)
    n = 1<<k
    raw = np.frombuffer(rng.randbytes(32*n), dtype=np.uint8).reshape(n,32).copy(); raw[:,0] &= 0x07
    f = ctx.ntt(raw); fo = O.ntt(raw)
    badf = np.nonzero((f != fo).any(axis=1))[0]
    i = ctx.ntt(raw, inverse=True); io = O.ntt(raw, inverse=True)
    badi = np.nonzero((i != io).any(axis=1))[0]
    print(k, "fwd bad", len(badf), badf[:8], "inv bad", len(badi), badi[:8], flush=True)
