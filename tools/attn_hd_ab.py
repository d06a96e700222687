import ctypes, os, sys, torch
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
def load(sfx):
    lib = ctypes.CDLL(os.path.join(ROOT, "clip_assisted_data_labeling_amd", "libclipenc_hip.so" if sfx == "cur" else f"libclipenc_hip_{sfx}.so"))
    f = lib.clipenc_op_attention
    f.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]; f.restype = ctypes.c_int
    return f
dev = torch.device("cuda", 0); crops, n_tok, heads, hd = 2048, 257, 16, int(os.environ.get('ATTN_HD', '80'))
T = crops * n_tok; width = heads * hd
qkv = (torch.randn(T, 3 * width, device=dev) * 1.5).to(torch.bfloat16)
st = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
o = torch.empty(T, width, device=dev, dtype=torch.bfloat16)
outs = {}
for n in sys.argv[1:]:
    f = load(n)
    for _ in range(2): assert f(qkv.data_ptr(), o.data_ptr(), crops, n_tok, width, heads, st) == 0
    torch.cuda.synchronize(); outs[n] = o.clone()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True); s.record()
    for _ in range(10): f(qkv.data_ptr(), o.data_ptr(), crops, n_tok, width, heads, st)
    e.record(); torch.cuda.synchronize(); ms = s.elapsed_time(e) / 10
    print(n, f"{ms:.3f} ms  {4.0 * crops * heads * n_tok * n_tok * hd / ms / 1e9:.0f} TFLOP/s", "equal to first:", bool(torch.equal(outs[n], outs[sys.argv[1]])),
          f"max |d| {(outs[n].float() - outs[sys.argv[1]].float()).abs().max().item():.3e}", "finite:", bool(torch.isfinite(outs[n].float()).all()))
