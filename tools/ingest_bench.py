import time, numpy as np, sys
sys.path.insert(0, '.')
from utmos_amd import device
n_var, n_samp = 2_000_000, 2504
rng = np.random.default_rng(0)
rows = rng.integers(0, 256, size=(n_var, (n_samp + 7)//8), dtype=np.uint8)
rows &= rng.integers(0, 256, size=rows.shape, dtype=np.uint8)
rows &= rng.integers(0, 256, size=rows.shape, dtype=np.uint8)   # ~12.5% density
with device.DeviceMatrix(n_samp) as m:
    c = m.add_chunk(n_var)
    t = time.perf_counter(); m.upload_rows_packed(c, rows); dt = time.perf_counter() - t
    print(f"upload_rows_packed: {rows.nbytes/1e6:.0f} MB in {dt*1e3:.0f} ms = {rows.nbytes/dt/1e9:.2f} GB/s")
    t = time.perf_counter(); m.upload_rows_packed(c, rows); dt = time.perf_counter() - t
    print(f"second time:        {rows.nbytes/dt/1e9:.2f} GB/s")
    t = time.perf_counter(); vc = m.var_count(); dt = time.perf_counter() - t
    print(f"var_count {dt*1e3:.1f} ms; check", int(vc[:3].sum()), int(np.unpackbits(rows[:, :1], axis=1)[:, :3].sum()))
    cols = m.download_columns(c, 0, 8)
    bits = np.unpackbits(rows[:, :1], axis=1)            # samples 0..7, MSB first
    exp = np.packbits(np.ascontiguousarray(bits.T), axis=1, bitorder='little').view('<u8')
    print('transpose ok', (cols[:, :exp.shape[1]] == exp).all())
    t = time.perf_counter(); m.upload_columns(c, m.download_columns(c)); dt = time.perf_counter() - t
    print(f"download+upload columns {dt*1e3:.0f} ms")
