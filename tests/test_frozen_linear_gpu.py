"""Frozen projections whose backward GEMM runs through a transposed weight copy (dalm_amd/models/frozen_linear.py): the same
function as `nn.Linear` forward and backward (the towers the reference runs through `self.generator_model(...)`,
dalm/models/rag_e2e_base_model.py:104-106, under its LoRA configuration :61-80 where every base projection is frozen)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    return torch.device("cuda:0")


def _rel(a, b):
    return float((a.double() - b.double()).norm() / (b.double().norm() + 1e-30))


@pytest.mark.parametrize("mode", ["bf16", "f32-autocast", "f32"])
@pytest.mark.parametrize("bias", [False, True])
def test_forward_equal_and_backward_close_to_nn_linear(dev, mode, bias):
    from dalm_amd.models import frozen_linear as FL

    torch.manual_seed(0)
    wdt = torch.bfloat16 if mode == "bf16" else torch.float32
    ref = torch.nn.Linear(512, 384, bias=bias).to(dev, wdt).requires_grad_(False)
    seq = torch.nn.Sequential(torch.nn.Linear(512, 384, bias=bias).to(dev, wdt).requires_grad_(False))
    seq[0].load_state_dict(ref.state_dict())
    assert FL.use_transposed_dgrad(seq) == 1 and type(seq[0]) is FL.FrozenLinearT
    x = torch.randn(6, 50, 512, device=dev, dtype=torch.float32 if mode != "bf16" else torch.bfloat16)
    up = torch.randn(6, 50, 384, device=dev)
    outs = []
    for m in (ref, seq[0]):
        xi = x.clone().requires_grad_(True)
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=mode == "f32-autocast"):
            y = m(xi)
        (y.float() * up).sum().backward()
        outs.append((y.detach(), xi.grad))
    assert torch.equal(outs[0][0], outs[1][0])                       # the forward is the same library call
    tol = 1e-6 if mode == "f32" else 3e-3                            # another GEMM kernel: summation order only
    assert outs[0][1].dtype == outs[1][1].dtype and _rel(outs[1][1], outs[0][1]) < tol
    wt = FL.dgrad_weight(seq[0].weight, torch.bfloat16 if mode != "f32" else torch.float32)
    assert wt.shape == (512, 384) and wt.is_contiguous() and torch.equal(wt.t(), seq[0].weight.to(wt.dtype))
    # the copy follows the weight
    with torch.no_grad():
        seq[0].weight.mul_(2.0)
    wt2 = FL.dgrad_weight(seq[0].weight, wt.dtype)
    assert torch.equal(wt2.t(), seq[0].weight.to(wt.dtype)) and not torch.equal(wt2, wt)


def test_pair_node_accumulates_one_dx_and_trainable_layers_keep_autograd(dev):
    from dalm_amd.models import frozen_linear as FL

    torch.manual_seed(1)
    m0 = torch.nn.Linear(256, 320, bias=False).to(dev, torch.bfloat16).requires_grad_(False)
    m1 = torch.nn.Linear(256, 320, bias=False).to(dev, torch.bfloat16).requires_grad_(False)
    x = torch.randn(40, 256, device=dev, dtype=torch.bfloat16)
    u0, u1 = torch.randn(40, 320, device=dev), torch.randn(40, 320, device=dev)
    xa = x.clone().requires_grad_(True)
    ((m0(xa).float() * u0).sum() + (m1(xa).float() * u1).sum()).backward()
    xb = x.clone().requires_grad_(True)
    y0, y1 = FL.pair_forward(xb, m0, m1)
    assert y0.grad_fn is y1.grad_fn and "Pair" in type(y0.grad_fn).__name__       # _FrozenPairFn, or _FrozenCatPairFn (one forward GEMM)
    ((y0.float() * u0).sum() + (y1.float() * u1).sum()).backward()
    assert _rel(xb.grad, xa.grad) < 4e-3
    # only one of the two outputs used: the other arrives as None
    xc = x.clone().requires_grad_(True)
    y0, y1 = FL.pair_forward(xc, m0, m1)
    (y1.float() * u1).sum().backward()
    xd = x.clone().requires_grad_(True)
    (m1(xd).float() * u1).sum().backward()
    assert _rel(xc.grad, xd.grad) < 4e-3
    # a trainable layer: plain nn.Linear semantics, weight gradient included
    m1.requires_grad_(True)
    xe = x.clone().requires_grad_(True)
    y0, y1 = FL.pair_forward(xe, m0, m1)
    assert y0.grad_fn is not y1.grad_fn
    (y0.float().sum() + y1.float().sum()).backward()
    assert m1.weight.grad is not None


@pytest.mark.parametrize("mode", ["bf16", "f32-autocast"])
def test_gate_up_as_one_gemm_equals_the_two_gemms(dev, mode):
    """`pair_forward` on frozen projections: ONE forward GEMM over [W0; W1] (the weights become views of one tensor, values and
    state_dict unchanged), SwiGLU reading / writing the halves in place (`dalm_swiglu_*_2d`): same values as the separate path."""
    from dalm_amd.models import frozen_linear as FL
    from dalm_amd.models import tower_ops

    torch.manual_seed(2)
    wdt = torch.bfloat16 if mode == "bf16" else torch.float32
    K, N, R = 512, 1408, 300
    m0 = torch.nn.Linear(K, N, bias=False).to(dev, wdt).requires_grad_(False)
    m1 = torch.nn.Linear(K, N, bias=False).to(dev, wdt).requires_grad_(False)
    sd0, sd1 = m0.weight.detach().clone(), m1.weight.detach().clone()
    x = torch.randn(2, R // 2, K, device=dev, dtype=torch.bfloat16 if mode == "bf16" else torch.float32)
    up = torch.randn(2, R // 2, N, device=dev)

    def run(cat: bool):
        old = FL._CAT
        FL._CAT = cat
        try:
            xi = x.clone().requires_grad_(True)
            with torch.autocast("cuda", dtype=torch.bfloat16, enabled=mode == "f32-autocast"):
                g, u = FL.pair_forward(xi, m0, m1)
                name = type(g.grad_fn).__name__
                act = tower_ops.swiglu(g, u)
            (act.float() * up).sum().backward()
            return name, g.detach().clone(), u.detach().clone(), act.detach(), xi.grad
        finally:
            FL._CAT = old

    sep = run(False)
    cat = run(True)
    assert "FrozenPairFn" in sep[0] and "FrozenCatPairFn" in cat[0]
    # the weights are now the halves of one tensor - same values, same keys
    assert torch.equal(m0.weight, sd0) and torch.equal(m1.weight, sd1)
    assert m1.weight.data_ptr() == m0.weight.data_ptr() + N * K * m0.weight.element_size()
    assert not m0.weight.requires_grad and m0.weight.is_contiguous() and m1.weight.is_contiguous()
    assert _rel(cat[1], sep[1]) < 2e-3 and _rel(cat[2], sep[2]) < 2e-3          # another GEMM solution: summation order only
    assert _rel(cat[3], sep[3]) < 4e-3 and _rel(cat[4], sep[4]) < 6e-3
    # SwiGLU on the halves EQUALS SwiGLU on contiguous copies of the same values (same arithmetic, strided addressing)
    g, u = cat[1], cat[2]
    both = torch.cat((g, u), dim=-1)
    gh, uh = both[..., :N].requires_grad_(True), both[..., N:].requires_grad_(True)
    assert tower_ops._halves_of_one(gh, uh)
    gc, uc = g.clone().requires_grad_(True), u.clone().requires_grad_(True)
    d = torch.randn_like(g)
    a_h = tower_ops.swiglu(gh, uh)
    a_c = tower_ops.swiglu(gc, uc)
    dg_h, du_h = torch.autograd.grad(a_h, (gh, uh), d)
    dg_c, du_c = torch.autograd.grad(a_c, (gc, uc), d)
    assert torch.equal(a_h, a_c) and torch.equal(dg_h, dg_c) and torch.equal(du_h, du_c)
    assert dg_h.is_contiguous() and du_h.is_contiguous()                          # the dgrad GEMMs keep their tuned shapes
    # a moved module: the cat is rebuilt, not used stale
    m0.to(torch.device("cpu")); m0.to(dev)
    xi = x.clone().requires_grad_(True)
    with torch.autocast("cuda", dtype=torch.bfloat16, enabled=mode == "f32-autocast"):
        g2, _ = FL.pair_forward(xi, m0, m1)
    assert _rel(g2, sep[1]) < 2e-3 and m1.weight.data_ptr() == m0.weight.data_ptr() + N * K * m0.weight.element_size()


def test_uncat_weights_restores_separate_storages(dev):
    from dalm_amd.models import frozen_linear as FL

    torch.manual_seed(3)
    mlp = torch.nn.ModuleDict({"gate_proj": torch.nn.Linear(128, 256, bias=False), "up_proj": torch.nn.Linear(128, 256, bias=False)})
    mlp = mlp.to(dev, torch.bfloat16).requires_grad_(False)
    want = {k: v.detach().clone() for k, v in mlp.state_dict().items()}
    x = torch.randn(8, 128, device=dev, dtype=torch.bfloat16, requires_grad=True)
    FL.pair_forward(x, mlp["gate_proj"], mlp["up_proj"])
    g, u = mlp["gate_proj"].weight, mlp["up_proj"].weight
    assert g.untyped_storage().data_ptr() == u.untyped_storage().data_ptr()
    assert FL.uncat_weights(mlp) == 1
    g, u = mlp["gate_proj"].weight, mlp["up_proj"].weight
    assert g.untyped_storage().data_ptr() != u.untyped_storage().data_ptr()
    assert all(torch.equal(v, want[k]) for k, v in mlp.state_dict().items())
    y0, y1 = FL.pair_forward(x, mlp["gate_proj"], mlp["up_proj"])                  # concatenates again
    assert mlp["gate_proj"].weight.untyped_storage().data_ptr() == mlp["up_proj"].weight.untyped_storage().data_ptr()
