"""Issue rate of the VALU instructions the tile kernel is made of (cycles per wave64 instruction per SIMD, assuming 2.4 GHz)."""
import ctypes as C, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gtsam_points_amd as gpa
from gtsam_points_amd import _capi
lib = gpa.load()
p = torch.zeros((1024, 3), dtype=torch.float32, device="cuda"); c = torch.zeros((1024, 9), dtype=torch.float32, device="cuda")
names = ["v_fma_f64", "v_mul_f64", "v_add_f64", "v_cvt_f64_f32", "v_cvt_f32_f64", "v_cvt_f64_i32", "v_fma_f32", "v_add_f32", "v_mov_b32", "v_mul_lo_u32",
         "v_rcp_f64", "v_cvt_i32_f64", "v_pk_fma_f32", "v_cndmask_b32", "v_fmac_f64", "v_lshl_add_u64"]
blocks = 256 * 8 * 4   # 8 waves per SIMD resident, 4 rounds
for op, name in enumerate(names):
    ms = C.c_float(); best = 1e9
    for _ in range(3):
        _capi.check(_capi.load_tune().gp_debug_stream_bench(C.c_void_p(p.data_ptr()), C.c_void_p(c.data_ptr()), blocks, 100 + op, 5, C.byref(ms)), "bench"); best = min(best, ms.value)
    wave_instr_per_simd = blocks * 4 * 64 * 64 / 1024
    print(f"{name:16s} {best*1e3:9.1f} us  {best*1e-3*2.4e9/wave_instr_per_simd:6.2f} cycles/wave-instr @2.4GHz", flush=True)
