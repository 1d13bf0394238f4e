"""Helpers shared by the parity tests."""
import numpy as np

# what the tolerance-based tests actually measured (excluded shares, worst errors): printed by conftest's terminal summary
PARITY_NOTES = []


def parity_note(line):
    PARITY_NOTES.append(str(line))


# one record per compared gradient array: which criterion decided it, measured error next to the tolerance
# (conftest writes them to gpurun_out/parity_rows.jsonl; tools/parity_table.py turns them into the table of DESIGN.md 3)
PARITY_ROWS = []


def parity_row(**row):
    PARITY_ROWS.append(row)


def to_dev(a, dev):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(dev)


def to_np(t):
    return t if isinstance(t, np.ndarray) else t.detach().cpu().numpy()


from oracle.parity import dilate  # noqa: E402,F401  (shared with __graft_entry__.smoke())


def rel_err(a, b):
    """max |a-b| relative to the largest magnitude of the reference b."""
    b = np.asarray(b, dtype=np.float64)
    a = np.asarray(a, dtype=np.float64)
    scale = max(float(np.abs(b).max()), 1e-30)
    return float(np.abs(a - b).max() / scale)


def assert_close_masked(got, want, tol, knife=None, what=""):
    """|got-want| <= tol * max|want| everywhere except at knife-edge positions (where the
    reference's strict `-1 < x < 1` test, models/transform.py:129, sits within rounding of
    the decision boundary and a 1-ulp difference legitimately flips a pixel to/from zero)."""
    got = np.asarray(got, dtype=np.float64)
    want = np.asarray(want, dtype=np.float64)
    assert got.shape == want.shape, (what, got.shape, want.shape)
    assert np.isfinite(got).all(), "%s: non-finite values" % what
    scale = max(float(np.abs(want).max()), 1e-30)
    bad = np.abs(got - want) > tol * scale
    if knife is not None:
        bad &= ~np.broadcast_to(knife, bad.shape)
    assert not bad.any(), "%s: %d / %d elements off by more than %g (max rel err %g)" % (
        what, int(bad.sum()), bad.size, tol, float((np.abs(got - want) * ~np.broadcast_to(
            knife if knife is not None else np.zeros(1, bool), bad.shape)).max() / scale))
