import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import neuralrecon_w_amd as nw
dev = torch.device("cuda:0")
net = nw.SDFNetwork(d_in=3, d_out=257, d_hidden=256, n_layers=8, skip_in=(4,)).to(dev)
for nwg in (256, 1024):
    x = (torch.rand(128 * nwg, 3, device=dev) * 2 - 1)
    for _ in range(3): out = net.sdf(x, prec=nw.PREC_BF16)
    torch.cuda.synchronize()
    t = out.reshape(-1)[: nwg * 4].view(torch.int32).cpu().view(nwg, 4).long() & 0xffffffff
    st, en = t[:, 0], t[:, 1]
    base = int(st.min())
    st, en = (st - base) / 100.0, (en - base) / 100.0   # s_memrealtime: 100 MHz -> us
    print("== %d WGs: first start 0, last start %.1f us, first end %.1f, last end %.1f us; WG duration min/median/max %.1f/%.1f/%.1f us"
          % (nwg, float(st.max()), float(en.min()), float(en.max()), float((en - st).min()), float((en - st).median()), float((en - st).max())))
    xcc = t[:, 3] & 0xf
    for k in range(8):
        m = xcc == k
        if m.any(): print("   xcc %d: %d WGs, start %.1f..%.1f, end %.1f..%.1f" % (k, int(m.sum()), float(st[m].min()), float(st[m].max()), float(en[m].min()), float(en[m].max())))
    order = torch.argsort(st)
    print("   start times (sorted, every 32nd):", " ".join("%.1f" % float(st[i]) for i in order[::32]))
