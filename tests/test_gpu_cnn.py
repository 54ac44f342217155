"""GPU parity tests (-m gpu) of the Nature-CNN dueling Q-network forward (BASELINE configs[4], PongNoFrameskip-v4 shape;
SURVEY.md 8(f) rank 4 -- not in the reference, PARITY UNPINNED): the implicit-GEMM kernels of dqn_cnn.hip, called through
the C ABI (dqn_cnn_*), against the CPU restatement (oracle/dqn_oracle_cnn.c, oracle_np.cnn_forward).

Bars: exact-f32 mode bit-identical to the C restatement's fmaf chains and within 1e-5 of f64; bf16 mode within 2e-2 of the
output scale; compute_q_targets (q_learning_functions.py:42-64) on top of it with the reference's terminal / one-hot rule."""
import numpy as np
import pytest

import _oracle as oc
from _oracle import onp

pytestmark = pytest.mark.gpu
A = 6            # Pong's action count


def host(t):
    return t.detach().cpu().numpy()


@pytest.fixture(scope="module")
def dq(torch_cuda):
    import deep_q_learning_amd as pkg
    return pkg


def make_params(seed):
    rng = np.random.default_rng(seed + 50)
    P = onp.cnn_init_params(A, seed)
    P = (P + 0.01 * rng.standard_normal(P.size)).astype(np.float32)       # non-zero biases everywhere
    return P


@pytest.mark.parametrize("B", [1, 5, 37])
def test_cnn_forward_f32_exact(dq, B):
    """exact-f32 MFMA mode: every conv / fc output is the k-ascending fmaf chain of the restatement -> identical bits; ragged
    last row tile of every layer (B * 400, B * 81, B * 49, B not multiples of the 64 / 128-row tiles)"""
    e = dq.CnnEngine(num_actions=A, max_batch=64, precision="f32")
    P, Pt = make_params(1), make_params(2)
    e.set_params(P); e.set_params(Pt, target=True)
    frames = np.random.default_rng(3 + B).integers(0, 256, (B, 84, 84, 4), dtype=np.uint8)
    q = host(e.forward(frames)); qt = host(e.forward(frames, target=True))
    qc, _ = oc.cnn_forward(P, frames, A)
    assert np.array_equal(q, qc), np.max(np.abs(q - qc))
    assert np.array_equal(qt, oc.cnn_forward(Pt, frames, A)[0])
    q64 = onp.cnn_forward(P, frames, A, np.float64)
    assert np.allclose(q, q64, rtol=1e-5, atol=1e-5)
    e.close()


def test_cnn_forward_bf16(dq):
    """bf16 MFMA mode (v_mfma_f32_32x32x16_bf16, f32 accumulate): 2e-2 of the output scale against f64; deterministic"""
    e = dq.CnnEngine(num_actions=A, max_batch=64, precision="bf16")
    P = make_params(4)
    e.set_params(P)
    frames = np.random.default_rng(5).integers(0, 256, (33, 84, 84, 4), dtype=np.uint8)
    q = host(e.forward(frames))
    q64 = onp.cnn_forward(P, frames, A, np.float64)
    scale = np.abs(q64).max()
    assert np.max(np.abs(q - q64)) <= 2e-2 * scale, (np.max(np.abs(q - q64)), scale)
    assert np.array_equal(q, host(e.forward(frames)))
    # not silently the f32 path
    assert np.max(np.abs(q - q64)) > 1e-6 * scale
    e.close()


def test_cnn_q_targets(dq):
    """compute_q_targets (q_learning_functions.py:42-64) with the CNN as the model: three forwards + the TD rule, incl.
    the terminal quirk Q3 (target of the taken action = q + r) and Q4 (q + delta * one_hot)"""
    B = 24
    e = dq.CnnEngine(num_actions=A, max_batch=32, precision="f32")
    P, Pt = make_params(6), make_params(7)
    e.set_params(P); e.set_params(Pt, target=True)
    rng = np.random.default_rng(8)
    s = rng.integers(0, 256, (B, 84, 84, 4), dtype=np.uint8); s2 = rng.integers(0, 256, (B, 84, 84, 4), dtype=np.uint8)
    a = rng.integers(0, A, B).astype(np.int32); r = rng.standard_normal(B).astype(np.float32); d = (rng.random(B) < 0.3).astype(np.float32)
    t = host(e.q_targets(s, a, r, s2, d, 0.99))
    q, _ = oc.cnn_forward(P, s, A); nq, _ = oc.cnn_forward(P, s2, A); nt, _ = oc.cnn_forward(Pt, s2, A)
    astar = nq.argmax(1)
    i = np.arange(B)
    t1 = np.float32(0.99) * nt[i, astar]; t2 = t1 - q[i, a]; t3 = (np.float32(1.0) - d) * t2; delta = r + t3      # :58
    want = q.copy(); want[i, a] = q[i, a] + delta                                                                 # :59
    assert np.array_equal(t, want)
    term = d > 0
    assert term.sum() >= 3 and np.array_equal(t[i, a][term], (q[i, a] + r)[term])                                 # quirk Q3
    e.close()
