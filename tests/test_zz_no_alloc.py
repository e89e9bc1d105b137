"""The entry point that allocates nothing (jirf_op_ln_conv1x1_static) against the entry point it is a
variant of (jirf_op_ln_conv1x1), through the C ABI.  With the allocating entry point on the fp32 matrix
instruction (jirf_set_gemm_split_bf16(0): the static one packs fp32 weights only, it has no room for the
three bf16 planes of the split form) both run the same kernels on the same packed weights, so the outputs
must be bit-identical; against the split form the two agree to fp32 rounding.  The static one keeps the
packed weights and the LayerNorm statistics in the library's per-GPU device array, which keeps its
contents between calls — hence the call pairs below, the second of each with a smaller problem."""
import numpy as np
import pytest
import torch

import jirf_hip
from tests.gpu_util import call, dev, hptr, ptr, stream

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def fp32_matrix_instruction(request):
    """Every test here but the one that names the split form compares like with like."""
    if "split" in request.node.name:
        yield
        return
    before = jirf_hip.lib().jirf_set_gemm_split_bf16(0)
    yield
    jirf_hip.lib().jirf_set_gemm_split_bf16(before)


def _both(kind, B, C, Cout, P, seed):
    rng = np.random.default_rng(seed)
    x = (rng.standard_normal((B, C, P)) * 2 + 0.5).astype(np.float32)
    lw = (1 + 0.1 * rng.standard_normal(C)).astype(np.float32)
    lb = (0.1 * rng.standard_normal(C)).astype(np.float32)
    w = (rng.standard_normal((Cout, C)) * C ** -0.5).astype(np.float32)
    xd, lwd, lbd, wd = dev(x), dev(lw), dev(lb), dev(w)
    want = torch.full((B, Cout, P), float("nan"), device="cuda")
    got = torch.full((B, Cout, P), float("nan"), device="cuda")
    call("jirf_op_ln_conv1x1", ptr(xd), ptr(lwd), ptr(lbd), kind, hptr(w), ptr(want), B, C, Cout, P, stream())
    call("jirf_op_ln_conv1x1_static", ptr(xd), ptr(lwd), ptr(lbd), kind, ptr(wd), ptr(got), B, C, Cout, P, stream())
    return got.cpu().numpy(), want.cpu().numpy()


@pytest.mark.parametrize("kind", [1, 2])
@pytest.mark.parametrize("first,second", [
    # Cin > 256: statistics pass + K-loop GEMM, mu / rstd in the static array
    ((2, 384, 1020, 4096), (1, 320, 254, 130)),
    # Cin <= 256: statistics inside the GEMM tile; only the packed weights use the array
    ((2, 96, 510, 4096), (1, 48, 144, 1000)),
    # a large pixel count first, then a small one (stale statistics beyond B * P must not matter)
    ((1, 288, 96, 256 * 256), (3, 272, 40, 77)),
])
def test_static_entry_point_matches_and_survives_a_smaller_second_call(kind, first, second):
    for i, (B, C, Cout, P) in enumerate((first, second)):
        got, want = _both(kind, B, C, Cout, P, seed=7 + i)
        assert np.isfinite(want).all()
        assert np.array_equal(got, want), f"call {i}: max |diff| = {np.abs(got - want).max()}"


@pytest.mark.parametrize("shape", [(2, 384, 1020, 4096), (2, 96, 510, 4096), (1, 192, 576, 16384)])
def test_static_entry_point_agrees_with_the_split_form(shape):
    """The allocating entry point on its default arithmetic (three bf16 terms per operand, six exact partial
    products, fp32 accumulation) against the static one (fp32 matrix instruction): fp32 rounding apart."""
    before = jirf_hip.lib().jirf_set_gemm_split_bf16(1)
    try:
        got, want = _both(1, *shape, seed=5)
    finally:
        jirf_hip.lib().jirf_set_gemm_split_bf16(before)
    assert np.isfinite(want).all()
    assert np.abs(got - want).max() < 2e-5, np.abs(got - want).max()   # |y| ~ 5: a few units in the last place


def test_static_entry_point_rejects_sizes_beyond_its_array():
    L = jirf_hip.lib()
    x = torch.zeros(8, device="cuda")
    # packed weights: ceil16(1024) * ceil128(1024) = 1 Mi floats > 512 Ki
    rc = L.jirf_op_ln_conv1x1_static(ptr(x), ptr(x), ptr(x), 1, ptr(x), ptr(x), 1, 1024, 1024, 64, stream())
    assert rc == -1 and "static scratch holds" in jirf_hip.last_error()
    # statistics: B * P beyond 1 Mi pixels
    rc = L.jirf_op_ln_conv1x1_static(ptr(x), ptr(x), ptr(x), 1, ptr(x), ptr(x), 5, 48, 48, 256 * 1024, stream())
    assert rc == -1 and "pixels" in jirf_hip.last_error()
    torch.cuda.synchronize()


def _operands(kind, B, C, Cout, P, seed):
    rng = np.random.default_rng(seed)
    x = (rng.standard_normal((B, C, P)) * 2 + 0.5).astype(np.float32)
    lw = (1 + 0.1 * rng.standard_normal(C)).astype(np.float32)
    lb = (0.1 * rng.standard_normal(C)).astype(np.float32)
    w = (rng.standard_normal((Cout, C)) * C ** -0.5).astype(np.float32)
    return w, dev(x), dev(lw), dev(lb), dev(w)


@pytest.mark.parametrize("shape", [(2, 384, 1020, 4096), (2, 96, 510, 4096)])
def test_static_entry_point_leaves_the_devices_free_memory_unchanged(shape):
    """What the name promises, measured: hipMemGetInfo before and after a call reports the same free memory (the
    packed weights and the statistics go to the library's static device array), while the entry point it is a
    variant of allocates and frees inside the call.  One warm-up call first: the first call on a device looks the
    array's address up and may load the code object."""
    L = jirf_hip.lib()
    B, C, Cout, P = shape
    w, xd, lwd, lbd, wd = _operands(1, B, C, Cout, P, 3)
    got = torch.empty((B, Cout, P), device="cuda")
    args = (ptr(xd), ptr(lwd), ptr(lbd), 1, ptr(wd), ptr(got), B, C, Cout, P, stream())
    call("jirf_op_ln_conv1x1_static", *args)
    torch.cuda.synchronize()
    free_before, _ = torch.cuda.mem_get_info()
    lows = []
    for _ in range(3):
        jirf_hip.check(L.jirf_op_ln_conv1x1_static(*args))
        lows.append(torch.cuda.mem_get_info()[0])          # queried while the call's kernels may still run
    torch.cuda.synchronize()
    free_after, _ = torch.cuda.mem_get_info()
    assert free_after == free_before and all(v == free_before for v in lows), (free_before, lows, free_after)


@pytest.mark.parametrize("kind", [1, 2])
@pytest.mark.parametrize("first,second", [((2, 384, 1020, 4096), (1, 320, 254, 130)),
                                          ((2, 96, 510, 4096), (1, 48, 144, 1000))])
def test_static_entry_point_inside_a_stream_capture(kind, first, second):
    """The callers it exists for: the second call of each pair is recorded into a hipGraph (no allocation, no
    synchronisation inside the call, so the capture stays valid) and replayed twice on new input data; the output
    equals the allocating entry point's bit for bit.  The first call of the pair runs outside the capture, which is
    the documented requirement: one call per device before any capture."""
    got, want = _both(kind, *first, seed=21)
    assert np.array_equal(got, want)
    B, C, Cout, P = second
    w, xd, lwd, lbd, wd = _operands(kind, B, C, Cout, P, 22)
    want = torch.full((B, Cout, P), float("nan"), device="cuda")
    got = torch.full((B, Cout, P), float("nan"), device="cuda")
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        jirf_hip.check(jirf_hip.lib().jirf_op_ln_conv1x1_static(ptr(xd), ptr(lwd), ptr(lbd), kind, ptr(wd), ptr(got), B, C,
                                                                Cout, P, stream()))
    for rep in range(2):
        xd.copy_(xd * 0.5 + 0.25 * rep)                    # the recorded call reads its operands at replay time
        graph.replay()
        torch.cuda.synchronize()
        call("jirf_op_ln_conv1x1", ptr(xd), ptr(lwd), ptr(lbd), kind, hptr(w), ptr(want), B, C, Cout, P, stream())
        assert np.array_equal(got.cpu().numpy(), want.cpu().numpy()), f"replay {rep}"
