import sys, time
import numpy as np
sys.path.insert(0, '.')
from utmos_amd import device
def build(chunk_vars, n_var, n_samp=2504, seed=0):
    m = device.DeviceMatrix(n_samp)
    v0 = 0
    while v0 < n_var:
        nv = min(chunk_vars, n_var - v0)
        c = m.add_chunk(nv); m.synth_fill(c, seed=seed, first_var_global=v0); v0 += nv
    return m
n_var = int(sys.argv[1])
res = {}
for cv in [int(x) for x in sys.argv[2:]]:
    m = build(cv, n_var); vc = m.var_count(); r = m.run(6); m.close()
    res[cv] = (vc, r)
    print(cv, 'vc sum', int(vc.sum()), 'rows', r[0].tolist(), r[1].tolist())
keys = list(res)
for k in keys[1:]:
    print('vc equal', keys[0], k, (res[keys[0]][0] == res[k][0]).all(), 'rows equal', (res[keys[0]][1][0] == res[k][1][0]).all() and (res[keys[0]][1][1] == res[k][1][1]).all())
