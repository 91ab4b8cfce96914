"""Host logic of trainer.FlatParams / the flat Adam step: same arithmetic as the stock per-tensor recipe
(Adam eps 1e-7 + global-norm clip 0.99, train.py:61, utils/__init__.py:23-31), unchanged state_dict."""
import copy

import torch


class _Net(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.l0, self.l1 = torch.nn.Linear(7, 16), torch.nn.Linear(16, 5)
        self.dead = torch.nn.Linear(3, 3)  # never used: like NeuconW.xyz_encoding_final / NeRF.views_linears

    def forward(self, x):
        return self.l1(torch.nn.functional.softplus(self.l0(x), beta=100))


def _net(seed):
    torch.manual_seed(seed)
    return _Net(), torch.nn.Embedding(11, 4)


def _loss(m, e, x, idx):
    return (m(x) ** 2).mean() + (e(idx) ** 2).sum() * 0.1


def test_flat_adam_matches_stock_recipe():
    from neuralrecon_w_amd.trainer import FlatParams

    m0, e0 = _net(0)
    m1, e1 = copy.deepcopy(m0), copy.deepcopy(e0)
    keys = list(m1.state_dict()) + list(e1.state_dict())
    g = torch.Generator().manual_seed(1)
    xs = [torch.randn(32, 7, generator=g) * 3 for _ in range(4)]
    idx = torch.tensor([1, 3, 3, 7])
    # stock: per-tensor Adam + clip_grad_norm_
    p0 = list(m0.parameters()) + list(e0.parameters())
    opt0 = torch.optim.Adam(p0, lr=1e-2, eps=1e-7)
    for x in xs:
        opt0.zero_grad(set_to_none=True)
        _loss(m0, e0, x, idx).backward()
        torch.nn.utils.clip_grad_norm_(p0, 0.99)
        opt0.step()
    # flat
    fp = FlatParams([e1, m1])
    assert list(m1.state_dict()) + list(e1.state_dict()) == keys
    assert all(isinstance(p, torch.nn.Parameter) for p in m1.parameters())
    opt1 = torch.optim.Adam([fp.flat], lr=1e-2, eps=1e-7)
    dead_before = m1.dead.weight.detach().clone()
    for x in xs:
        fp.zero_grad()
        _loss(m1, e1, x, idx).backward()
        for p in fp.params:  # gradients were accumulated IN PLACE into the flat buffer
            off, k = fp.slices[id(p)]
            assert p.grad.data_ptr() == fp.flat_grad[off:off + k].data_ptr()
        torch.nn.utils.clip_grad_norm_([fp.flat], 0.99)
        opt1.step()
    for (k0, v0), (k1, v1) in zip(list(m0.state_dict().items()) + list(e0.state_dict().items()),
                                  list(m1.state_dict().items()) + list(e1.state_dict().items())):
        assert k0 == k1
        assert torch.allclose(v0, v1, rtol=1e-6, atol=1e-7), k0
    assert torch.equal(m1.dead.weight, dead_before)  # zero gradient -> exactly zero Adam update


def test_flat_params_survive_set_to_none():
    from neuralrecon_w_amd.trainer import FlatParams

    m, e = _net(2)
    fp = FlatParams([m, e])
    for p in m.parameters():
        p.grad = None
    fp.zero_grad()
    _loss(m, e, torch.randn(8, 7), torch.tensor([0, 1])).backward()
    assert float(fp.flat_grad.abs().sum()) > 0
    off, k = fp.slices[id(m.l0.weight)]
    assert torch.equal(fp.flat_grad[off:off + k].view_as(m.l0.weight), m.l0.weight.grad)
