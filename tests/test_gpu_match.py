"""GPU parity: K7 cross-checked Hamming matcher + gate vs the CPU oracle (bit-exact indices and distances).
Reference path: VO::feature_matching, visual_odometry.cpp:219-251."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _same(a, b):
    assert len(a) == len(b), (len(a), len(b))
    for f in ("queryIdx", "trainIdx", "imgIdx", "distance"):
        assert (a[f] == b[f]).all(), f


@pytest.mark.parametrize("nq,nt,seed", [(500, 500, 1), (1500, 1500, 2), (137, 911, 3), (911, 137, 4), (1, 1, 5), (4096, 4096, 6),
                                        (64, 65, 7), (257, 255, 8)])
def test_match_parity_random(vo, oracle, synth, nq, nt, seed):
    q, t = synth.random_descriptors(nq, nt, seed=seed)
    for gate in (False, True):
        got = vo.feature_matching(q, t, 1.0, gate=gate)
        want = oracle.feature_matching(q, t, 1.0) if gate else oracle.bf_match_xcheck(q, t)
        _same(got, want)


def test_match_gate_frame_gap(vo, oracle, synth):
    q, t = synth.random_descriptors(600, 700, seed=11, flip_p=0.12)
    for gap in (1.0, 2.0, 3.0, 0.5):
        _same(vo.feature_matching(q, t, gap), oracle.feature_matching(q, t, gap))


def test_match_engineered_ties(vo, oracle):
    rng = np.random.default_rng(0)
    base = rng.integers(0, 256, (8, 32), dtype=np.uint8)
    q = np.concatenate([base, base, base[:3]])          # duplicated query rows
    t = np.concatenate([base[::-1], base, base[2:5]])   # duplicated train rows
    _same(vo.feature_matching(q, t, 1.0, gate=False), oracle.bf_match_xcheck(q, t))
    z = np.zeros((40, 32), np.uint8)                     # all distances equal (0)
    _same(vo.feature_matching(z, z, 1.0, gate=False), oracle.bf_match_xcheck(z, z))


def test_match_empty(vo):
    e = np.zeros((0, 32), np.uint8); one = np.zeros((3, 32), np.uint8)
    assert len(vo.feature_matching(e, one)) == 0
    assert len(vo.feature_matching(one, e)) == 0


def test_match_batched_dev(vo, oracle, synth):
    """device-resident batched call through torch tensors (the throughput path)"""
    import torch
    B, cap = 4, 1536
    qs, ts, nq, nt = [], [], [], []
    for b in range(B):
        n1, n2 = 900 + 150 * b, 1500 - 100 * b
        q, t = synth.random_descriptors(n1, n2, seed=40 + b)
        qq = np.zeros((cap, 32), np.uint8); qq[:n1] = q; tt = np.zeros((cap, 32), np.uint8); tt[:n2] = t
        qs.append(qq); ts.append(tt); nq.append(n1); nt.append(n2)
    dev = torch.device("cuda:0")
    dq = torch.from_numpy(np.stack(qs)).to(dev); dt = torch.from_numpy(np.stack(ts)).to(dev)
    dnq = torch.tensor(nq, dtype=torch.int32, device=dev); dnt = torch.tensor(nt, dtype=torch.int32, device=dev)
    dgap = torch.ones(B, dtype=torch.float64, device=dev)
    dout = torch.zeros((B, cap, 16), dtype=torch.uint8, device=dev); dn = torch.zeros(B, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    vo.feature_matching_dev(dq.data_ptr(), cap * 32, dnq.data_ptr(), dt.data_ptr(), cap * 32, dnt.data_ptr(), dgap.data_ptr(), 1, B,
                            cap, dout.data_ptr(), cap, dn.data_ptr())
    vo.sync()
    out = dout.cpu().numpy(); n = dn.cpu().numpy()
    from stereo_visual_slam_amd import DMATCH_DTYPE
    for b in range(B):
        got = out[b].reshape(-1).view(DMATCH_DTYPE)[:n[b]]
        _same(got, oracle.feature_matching(qs[b][:nq[b]], ts[b][:nt[b]], 1.0))
