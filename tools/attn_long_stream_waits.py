import ctypes, os, sys, torch
ROOT = "/root/repo" if not os.environ.get("GRAFT_REPO_ROOT") else os.environ["GRAFT_REPO_ROOT"]
lib = ctypes.CDLL(os.path.join(ROOT, "clip_assisted_data_labeling_amd", "libclipenc_hip_" + (sys.argv[1] if len(sys.argv) > 1 else "cnt") + ".so"))
f = lib.clipenc_op_attention
f.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
dev = torch.device("cuda", 0)
crops, n_tok = 480, 577
T = crops * n_tok
qkv = (torch.randn(T, 3072, device=dev) * 1.5).to(torch.bfloat16)
st = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
o = torch.zeros(T, 1024, device=dev, dtype=torch.bfloat16)
for _ in range(3):
    f(qkv.data_ptr(), o.data_ptr(), crops, n_tok, 1024, 16, st)
torch.cuda.synchronize()
v = o.view(torch.int64).flatten()[: 256 * 12 * 2].view(256, 12, 2).cpu()
wait, tot = v[:, :11, 0].float(), v[:, :11, 1].float()
print("compute waves: waiting for tiles / wave lifetime: mean %.3f  median %.3f  max %.3f" % ((wait / tot).mean(), (wait / tot).median(), (wait / tot).max()))
print("wave lifetime cycles: mean %.0f min %.0f max %.0f" % (tot.mean(), tot.min(), tot.max()))
print("per wave index mean wait share:", [(round(float((wait[:, i] / tot[:, i]).mean()), 3)) for i in range(11)])

flat = o.view(torch.int64).flatten().cpu()
ld = flat[8192: 8192 + 256 * 4].view(256, 4).float()
life = tot.mean()
print("loader: issuing %.3f  blocked on the oldest tile %.3f  idle (nothing free, nothing in flight) %.3f  of a compute wave's lifetime" % (ld[:, 0].mean() / life, ld[:, 1].mean() / life, ld[:, 2].mean() / life))
w = flat[16384: 16384 + 256 * 12 * 2].view(256, 12, 2)[:, :11].float()
print("consumer wait share at tiles 0-1: %.3f, at the last 4 tiles: %.3f (of all waiting)" % (w[:, :, 0].sum() / wait.sum(), w[:, :, 1].sum() / wait.sum()))
