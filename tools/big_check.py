"""One-off check at BASELINE configs[4] scale (500M variants x 2,504 samples in 10 HBM-resident chunks, 156 GB):
first iterations brute force vs decremental, invariants, and the same rows from a re-chunked layout."""
import sys, time
import numpy as np
sys.path.insert(0, '.')
from utmos_amd import device

def build(chunk_vars, n_var=500_000_000, n_samp=2504, seed=0):
    m = device.DeviceMatrix(n_samp)
    v0 = 0
    while v0 < n_var:
        nv = min(chunk_vars, n_var - v0)
        c = m.add_chunk(nv)
        m.synth_fill(c, seed=seed, first_var_global=v0)
        v0 += nv
    return m

k = 24
t = time.time(); m = build(50_000_000); print('built 10 chunks', round(time.time() - t, 1), 's')
vc = m.var_count()
a = m.run(k); st = m.stats()
print('brute force', a[0][:6], a[1][:6], 'GB/s', st['algo_bytes'] / st['loop_ms'] / 1e6)
m.set_decremental(True, 1.0); m.reset(); b = m.run(k); st = m.stats()
print('decremental iterations', st['decr_iterations'])
print('a==b', (a[0] == b[0]).all(), (a[1] == b[1]).all(), 'monotone', (np.diff(a[1]) <= 0).all(), 'first', a[1][0], vc.max(), int(np.argmax(vc)), 'unique', len(set(a[0].tolist())))
assert (a[0] == b[0]).all() and (a[1] == b[1]).all()
assert (np.diff(a[1]) <= 0).all() and a[1][0] == vc.max() and len(set(a[0].tolist())) == k
m.close()
m = build(125_000_000); c = m.run(k); m.close()     # 4 chunks: same matrix, same rows
assert (a[0] == c[0]).all() and (a[1] == c[1]).all()
print('ok: brute force == decremental == re-chunked at 500M x 2504')
