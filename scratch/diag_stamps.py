import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "multimodal-framework-for-speaker-emotion-recognition_amd")]
import numpy as np, torch
from mser import ops
from models.lsthm_sps import MARN_cell
torch.manual_seed(0)
HID = int(sys.argv[1]) if len(sys.argv) > 1 else 128
NDIR = int(sys.argv[2]) if len(sys.argv) > 2 else 1          # 2: a second direction beside the stamped one (same tables; timing only)
m = MARN_cell(HID, HID, 100, 100, dh_s=HID).cuda()
T, N = 128, 32
rs = np.random.RandomState(0)
x_l = torch.tensor(rs.standard_normal((T, N, 100)).astype(np.float32)).cuda()
x_a = torch.tensor(rs.standard_normal((T, N, 100)).astype(np.float32)).cuda()
qmask = torch.tensor(np.eye(2, dtype=np.float32)[rs.randint(0, 2, (T, N))]).cuda()
P = dict(m.named_parameters())
G = {k: torch.zeros_like(v) for k, v in P.items()}
out = torch.zeros(T * N, 4 * HID, device="cuda"); dout = torch.randn(T * N, 4 * HID, device="cuda")
ws = torch.zeros(ops.cell_workspace_bytes(T, N, 100, HID, NDIR), device="cuda", dtype=torch.uint8)
out2 = torch.zeros(T * N, 4 * HID, device="cuda")
dirs = [dict(p=ops.cell_param_struct(lambda n: P[n].detach()), g=ops.cell_param_struct(lambda n: G[n]), qmask=qmask, rev=None, out=out, dout=dout)]
if NDIR == 2:
    G2 = {k: torch.zeros_like(v) for k, v in P.items()}
    rev = None
    if len(sys.argv) > 4 and sys.argv[4] == "rev":            # the second direction reversed in time (full lengths), as the model runs it
        rev = torch.arange(T - 1, -1, -1, dtype=torch.int32, device="cuda").view(T, 1).expand(T, N).contiguous()
    dirs.append(dict(p=ops.cell_param_struct(lambda n: P[n].detach()), g=ops.cell_param_struct(lambda n: G2[n]), qmask=qmask, rev=rev, out=out2, dout=dout))
dx_l, dx_a = torch.zeros(T * N, 100, device="cuda"), torch.zeros(T * N, 100, device="cuda")
desc = ops.make_cell_desc(T, N, 100, HID, x_l.view(T * N, 100), x_a.view(T * N, 100), dirs, 4 * HID, ws, dx_l=dx_l, dx_a=dx_a)
for _ in range(2):
    ops.marn_cell_fwd(desc)
ops.marn_cell_status(desc)
for _ in range(2):
    ops.marn_cell_bwd(desc)
ops.marn_cell_status(desc)
# phase stamps (build with `make EXTRA=-DMSER_STAMPS`): average s_memrealtime ticks (10 ns) per step and phase
w = ws.view(torch.int32).cpu().numpy()
SYNC_STAMPS = 8 * 8 * 32 + 10 * 32 + 32      # SYNC_ABORT + SYNC_LINE (recurrent.hip)
for name, base in (("lsthm_fwd [z loads(+poll), mm, epilogue, hq early mm (+barrier1), row (+c poll), h early mm (+barrier2)]", 24), ("lsthm_bwd [row tail, arrive|barrier1, mat rest, post-mat, (sv2: A poll, mfma), carries+coef, pass2]", 32),
                   ("spk_bwd wg0 [sync, product, epilogue, barrier, table loads, load issue, wait+compute+stores]", 40), ("spk_bwd wg(2,1)", 48)):
    v = w[SYNC_STAMPS - 16 + base: SYNC_STAMPS - 16 + base + 8]
    print(name, [round(int(x) * 0.01, 2) for x in v], "us/step; sum", round(float(v.sum()) * 0.01, 2))

# launch durations of the fused chain kernels in this cell-only setting (HIP events inside libmser): usage: ... <H> <ndir> time
if len(sys.argv) > 3 and sys.argv[3] == "time":
    import ctypes
    from mser import _lib as L_
    lib = L_.load()
    for nm, kid, fn in (("cell_fwd_fused", 2, ops.marn_cell_fwd), ("cell_bwd_fused", 4, ops.marn_cell_bwd)):
        nrep = int(sys.argv[5]) if len(sys.argv) > 5 else 5
        L_.check(lib.mser_prof_enable(kid, 2 * nrep + 64), "prof_enable")
        for _ in range(nrep):
            fn(desc)
        torch.cuda.synchronize()
        tot, cnt = ctypes.c_float(0), ctypes.c_int32(0)
        L_.check(lib.mser_prof_collect(ctypes.byref(tot), ctypes.byref(cnt)), "prof_collect")
        lib.mser_prof_enable(0, 0)
        print(f"{nm}: {tot.value * 1e3 / max(cnt.value, 1):.1f} us per launch ({cnt.value} launches), T={T} ndir={NDIR}")

# does the state of the caches matter?  usage: ... <H> <ndir> trash [rev]: preparation, then 1 GB of unrelated traffic, then the chain launch
if len(sys.argv) > 3 and sys.argv[3] == "trash":
    import ctypes
    from mser import _lib as L_
    lib = L_.load()
    big = torch.empty(256 * 1024 * 1024, device="cuda")
    for label, trash in (("prep -> chain", False), ("prep -> 1 GB fill -> chain", True)) * 2:
        L_.check(lib.mser_prof_enable(2, 256), "prof_enable")
        for _ in range(20):
            ops.marn_cell_run(desc, ops.PHASE_FWD_PREP | ops.PHASE_SPEAKER_FWD)
            if trash:
                big.fill_(1.0)
            ops.marn_cell_run(desc, ops.PHASE_LSTHM_FWD)
        torch.cuda.synchronize()
        tot, cnt = ctypes.c_float(0), ctypes.c_int32(0)
        L_.check(lib.mser_prof_collect(ctypes.byref(tot), ctypes.byref(cnt)), "prof_collect")
        lib.mser_prof_enable(0, 0)
        print(f"cell_fwd_fused, {label}: {tot.value * 1e3 / max(cnt.value, 1):.1f} us per launch ({cnt.value} launches)")
