import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tools"))
from eav_amd import _lib
import gemm_sp_bench as gb
_lib.load()
M, N, K = 9712, 3072, 768
A = torch.randn(M, K, device="cuda"); B = torch.randn(N, K, device="cuda") * 0.02
sa, pa, _ = gb.planes(A); sb, pb, _ = gb.planes(B)
C = torch.empty(M, N, device="cuda")
_lib.call("eav_gemm_sp_set_tile", 1)
for _ in range(3):
    _lib.call("eav_gemm_sp", pa.data_ptr(), pb.data_ptr(), C.data_ptr(), sa.data_ptr(), sb.data_ptr(), M, N, K, N, 1, 0, 0, 1.0, None, 0, None, None, 0, 0, None, None)
torch.cuda.synchronize()
