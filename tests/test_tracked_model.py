"""The arithmetic fact ODDIO_HIP_MODE_TRACKED rests on, on the CPU: a sequential f32 sum (the reference's `*o += s * gain`
over the sources, src/spatial.rs:204,459-460) rounds every addend to the ulp of the running sum's binade, so a running sum that is
restarted in the middle of the walk at (nearly) the value the reference's sum has there makes the same rounding errors as the
reference, and the differences end - start of such restarted blocks add up to the reference's result -- rounding errors included --
where an exact (tree) sum of the same addends is as far from it as the reference is from the exact sum.  numpy model of the device
scheme: block sums, their prefixes in walk order, restarted blocks, the sum of the differences."""
import numpy as np


def _model(n_sources, n_out, block, seed, perturb=0.0):
    rng = np.random.default_rng(seed)
    c = (rng.standard_normal((n_sources, n_out)) * 0.2).astype(np.float32)
    ref = np.zeros(n_out, np.float32)
    for s in range(n_sources):                                   # the reference: one running f32 sum per output
        ref = ref + c[s]
    exact = c.astype(np.float64).sum(0)
    nb = n_sources // block
    cb = c.reshape(nb, block, n_out)
    t = cb.astype(np.float64).sum(1).astype(np.float32)           # first pass: every block's sum (a tree on the device: ~exact)
    prefix = np.zeros((nb, n_out), np.float32)
    run = np.zeros(n_out, np.float32)
    for b in range(nb):                                          # track_prefix: where the walk stands when it reaches block b
        prefix[b] = run
        run = run + t[b]
    if perturb:                                                  # start values off by `perturb` x the output's peak, independently per block and output
        peak_ = float(np.abs(exact).max())
        prefix = (prefix + (rng.standard_normal(prefix.shape) * perturb * peak_).astype(np.float32)).astype(np.float32)
    acc = prefix.copy()
    for i in range(block):                                       # second pass: every block restarted at its prefix, all blocks at once
        acc = acc + cb[:, i, :]
    d = acc - prefix                                             # what each block added, as rounded at the reference's magnitudes
    x = d
    while x.shape[0] > 1:                                        # the reduce: an f32 tree over the blocks' differences
        if x.shape[0] & 1:
            x = np.concatenate([x, np.zeros((1, n_out), np.float32)])
        x = x[0::2] + x[1::2]
    tracked = x[0]
    tree = t.astype(np.float64).sum(0).astype(np.float32)
    peak = float(np.abs(exact).max())
    return (float(np.abs(tracked - ref).max()) / peak, float(np.abs(tree - ref).max()) / peak, float(np.abs(ref - exact).max()) / peak)


def test_restarted_blocks_repeat_the_sequential_sums_rounding_errors():
    e_tracked, e_tree, e_ref = _model(65536, 48, 128, 7)
    # the tree is as far from the reference as the reference is from the exact sum; the tracked sum is several times closer
    assert e_tree > 3e-6 and abs(e_tree - e_ref) < 0.5 * e_ref, (e_tracked, e_tree, e_ref)
    assert e_tracked < 1e-6 and e_tracked < 0.25 * e_tree, (e_tracked, e_tree, e_ref)


def test_small_blocks_and_ragged_block_counts():
    e_tracked, e_tree, _ = _model(16 * 1000, 16, 16, 11)        # 1000 blocks of one group of 16 sources
    assert e_tracked < 1e-6 and e_tracked < 0.5 * e_tree, (e_tracked, e_tree)


def test_start_values_only_have_to_place_the_binade():
    """What the first pass has to deliver: the restarted sums repeat the reference's rounding errors as long as
    they sit in the reference's binade, so start values off by 1e-3 of the output's peak -- four orders of magnitude more than the
    first pass's own error -- cost nothing; at 1e-2 the mismatched steps begin to show."""
    base = _model(65536, 48, 128, 7)[0]
    for perturb, bound in ((1e-5, 1e-6), (1e-4, 1e-6), (1e-3, 1e-6), (1e-2, 3e-6)):
        e = _model(65536, 48, 128, 7, perturb)[0]
        assert e < bound, (perturb, e, base)
