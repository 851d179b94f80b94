import sys
sys.path.insert(0, 'tools')
import trajectory as tj
for size in (64, 96):
    state = tj.start_state()
    x, t = tj.denoise_pairs(4, size, 11); xh, th = tj.denoise_pairs(2, size, 23)
    cps = (10, 25, 50)
    ref = tj.run_torch("float64", state, x, t, xh, th, 50, cps)
    print("size", size)
    for name, kw in (("float32", dict(dtype_name="float32")), ("f32 24-bit operands (control)", dict(dtype_name="float32", sig_bits=24)), ("f32 20-bit", dict(dtype_name="float32", sig_bits=20)),
                     ("f32 16-bit", dict(dtype_name="float32", sig_bits=16)), ("f32 12-bit", dict(dtype_name="float32", sig_bits=12)), ("f64 16-bit", dict(dtype_name="float64", sig_bits=16))):
        dn = kw.pop("dtype_name")
        r = tj.run_torch(dn, state, x, t, xh, th, 50, cps, **kw)
        print(f"  {name:32s}" + "  ".join(f"step {c}: dloss {abs(r[0][c-1]-ref[0][c-1]):.2e} dPSNR {max(abs(a-b) for a,b in zip(r[1][c], ref[1][c])):.2e}" for c in cps), flush=True)
    for m in tj.ENGINE_MODES:
        r = tj.run_engine(m, state, x, t, xh, th, 50, cps)
        print(f"  {m:32s}" + "  ".join(f"step {c}: dloss {abs(r[0][c-1]-ref[0][c-1]):.2e} dPSNR {max(abs(a-b) for a,b in zip(r[1][c], ref[1][c])):.2e}" for c in cps), flush=True)
