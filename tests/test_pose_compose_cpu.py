"""Host logic of the pose composition (warp.compose_pose; reference: modellearn_proj_center.py:388-404) on the CPU path: against the
rotation-matrix form q = q3 * q_prev, t = R(q3) t_prev + t3 for unit quaternions, and the chain-kernel error counter's host side."""
import torch


def _rot(q):
    w, x, y, z = q.unbind(-1)
    return torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y),
                        2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x),
                        2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)], -1).view(*q.shape[:-1], 3, 3)


def test_compose_pose_matches_rotation_matrix_form(oracle_backend):
    from i2pnet_amd import ops, warp
    prev = ops.set_backend(oracle_backend)
    try:
        g = torch.Generator().manual_seed(0)
        B = 6
        q3 = torch.nn.functional.normalize(torch.randn(B, 4, generator=g, dtype=torch.float64), dim=1).float()
        qp = torch.nn.functional.normalize(torch.randn(B, 4, generator=g, dtype=torch.float64), dim=1).float()
        t3, tp = torch.randn(B, 3, generator=g), torch.randn(B, 3, generator=g)
        out = warp.compose_pose(q3, t3, qp, torch.cat([torch.zeros(B, 1), tp], 1))
        assert out.shape == (B, 7)
        want_t = (_rot(q3) @ tp.unsqueeze(-1)).squeeze(-1) + t3
        assert torch.allclose(out[:, 4:], want_t, atol=2e-5)
        # the rotation of the composed quaternion is the product of the rotations
        assert torch.allclose(_rot(out[:, :4]), _rot(q3) @ _rot(qp), atol=2e-5)
    finally:
        ops.set_backend(prev)


def test_chain_error_counter_is_zero_without_launches():
    from i2pnet_amd import ops
    assert ops.chain_errors() == 0
