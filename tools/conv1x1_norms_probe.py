import sys, time, torch
sys.path.insert(0, ".")
from quantized_neural_networks_amd import hip
for (n,H,W,cin,sh) in [(4096,56,56,64,1),(4096,28,28,512,1),(4096,56,56,256,2)]:
    g = torch.Generator(device="cuda").manual_seed(2)
    act_q = torch.relu(torch.randn((n, H, W, cin), device="cuda", generator=g))
    best=[1e9,1e9]
    for it in range(3):
        torch.cuda.synchronize(); t0=time.time()
        planes = act_q.permute(3, 0, 1, 2)[:, :, ::sh, ::sh].reshape(cin, -1).contiguous()
        nrm = hip.row_norms(planes)
        torch.cuda.synchronize(); best[0]=min(best[0],time.time()-t0)
        del planes
        torch.cuda.synchronize(); t0=time.time()
        nrm2 = hip.channel_sumsq(act_q, (sh, sh)).sqrt().float()
        torch.cuda.synchronize(); best[1]=min(best[1],time.time()-t0)
    print(f"{cin} ch @{H}x{W}/{sh} n={n}: channel-major copy + row norms {best[0]*1e3:.2f} ms; channel_sumsq {best[1]*1e3:.2f} ms ({act_q.numel()*4/sh/sh/best[1]/1e12:.2f} TB/s); equal: {bool(torch.equal(nrm, nrm2))}")
    del act_q
