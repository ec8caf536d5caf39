"""The ctypes stub of INTEGRATION.md §2, run as is on the GPU box (binding the C ABI without this package's Python)."""
import ctypes, torch, os
os.chdir(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
lib = ctypes.CDLL("video_similarity_search_amd/csrc/libslic_hip.so")
P, I, L = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64
lib.slic_last_error.restype = ctypes.c_char_p
lib.slic_kmeans_cnorm.argtypes = [P, I, I, I, P, P]
lib.slic_kmeans_assign_workspace_bytes.restype = ctypes.c_size_t
lib.slic_kmeans_assign_workspace_bytes.argtypes = [L, I]
lib.slic_kmeans_assign.argtypes = [P, L, I, I, P, I, I, P, P, P, P, P, P, P]
def assign(X, C):
    N, D = X.shape; K = C.shape[0]
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    cn = torch.empty(K, device=X.device); lab = torch.empty(N, dtype=torch.int32, device=X.device)
    ws = torch.empty(lib.slic_kmeans_assign_workspace_bytes(N, K), dtype=torch.uint8, device=X.device)
    for rc in (lib.slic_kmeans_cnorm(C.data_ptr(), K, D, D, cn.data_ptr(), st),
               lib.slic_kmeans_assign(X.data_ptr(), N, D, D, C.data_ptr(), K, D, cn.data_ptr(), lab.data_ptr(),
                                      None, None, None, ws.data_ptr(), st)):
        if rc: raise RuntimeError(lib.slic_last_error().decode())
    return lab
X = torch.randn(1000, 64, device="cuda"); C = X[:10].clone()
lab = assign(X, C)
ref = ((X[:, None] - C[None]) ** 2).sum(-1).argmin(1)
print("stub ok", (lab.long() == ref).float().mean().item())
