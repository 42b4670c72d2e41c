"""CIN fwd+bwd step time against the batch size (c4 shape otherwise): how the kernels hold up when a strong-scaling run
leaves 512 samples per GPU.  Run on the GPU box:  python tools/cin_batch_sweep.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ml_function_amd import functional as Fn, synth

def main():
    F, K, conv = 39, 16, [128, 128, 128]
    for B in (256, 512, 1024, 2048, 4096, 8192):
        c = synth.cin_case(B, F, K, conv, dist="uniform")
        d = lambda a: torch.tensor(a, device="cuda")
        x = d(c["x"]).requires_grad_()
        Ws = [d(w).requires_grad_() for w in c["Ws"]]
        bs = [d(b).requires_grad_() for b in c["bs"]]
        dw, db = d(c["dense_w"]).requires_grad_(), d(c["dense_b"]).requires_grad_()
        g = d(c["g"])
        def step():
            out = Fn.cin(x, Ws, bs, dw, db, output_dim=1)
            out.backward(g)
        for _ in range(5):
            step()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 20
        e0.record()
        for _ in range(n):
            step()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / n
        from ml_function_amd import _lib
        _lib.profile_begin("cin_fwd_,cin_bwd_dw_,cin_bwd_dz_")
        for _ in range(5):
            step()
        torch.cuda.synchronize()
        pr = _lib.profile_end()
        print("B=%5d  %.3f ms/step  %.3f M samples/s  (%.2f us/sample)   " % (B, ms, B / ms / 1e3, ms * 1e3 / B)
              + "  ".join("%s %.3f" % (k.replace("cin_", ""), v["avg_ms"]) for k, v in sorted(pr.items())))

if __name__ == "__main__":
    main()
