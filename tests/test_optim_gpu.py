"""FlatAdam (csrc/optim.hip) follows torch.optim.Adam -- the optimizer of the reference's trainer (option_new.py:83-90)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("wd", [0.0, 0.01])
def test_flat_adam_matches_torch_adam(dev, wd):
    from gcanet_amd import parallel
    from gcanet_amd.optim import FlatAdam
    torch.manual_seed(0)
    shapes = [(64, 33), (7,), (128, 64, 1), (5, 3, 2), (1,)]        # total not a multiple of 4: exercises the tail
    ref = [torch.nn.Parameter(torch.randn(*s, device=dev)) for s in shapes]
    mod = torch.nn.ParameterList([torch.nn.Parameter(p.detach().clone()) for p in ref])
    dp = parallel.FlatGradDP(mod, 1)
    mine = FlatAdam(dp, lr=3e-3, betas=(0.9, 0.99), eps=1e-8, weight_decay=wd)
    theirs = torch.optim.Adam(ref, lr=3e-3, betas=(0.9, 0.99), eps=1e-8, weight_decay=wd)
    g = torch.Generator(device="cpu").manual_seed(1)
    for step in range(6):
        grads = [torch.randn(*s, generator=g).to(dev) * (10.0 ** (step - 3)) for s in shapes]
        dp.zero_grad()
        for p, q, gr in zip(mod, ref, grads):
            p.grad = gr.clone()
            q.grad = gr.clone()
        dp.all_reduce_grads()
        mine.step()
        theirs.step()
        for p, q in zip(mod, ref):
            assert torch.allclose(p, q, rtol=1e-6, atol=1e-7), (step, (p - q).abs().max().item())   # bias corrections in double, as torch
    assert float(mine.state[0]) == 6.0
    assert all(p.data_ptr() >= mine.flat_p.data_ptr() for p in mod)      # parameters are views of the flat buffer
