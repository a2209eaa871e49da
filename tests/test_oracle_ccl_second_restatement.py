"""A SECOND, independent restatement of OpenCV's label numbering for 8-connectivity, checked against the oracle's.

The reference calls cv::connectedComponentsWithStats(patch, ..., 8, CV_32S) (src/cont2/contour_mng.cpp:298) and inserts the
components into cont_views_ in label order, so the numbering decides the pre-sort order and with it the order among
equal-sized contours.  OpenCV is absent here.  oracle/orc_contour.h states the numbering as a derived RULE ("components
are numbered by their first 2x2 block in block-raster order relative to the patch origin", implemented as flood fill +
sort by that key).  This file restates the ALGORITHM the rule was derived from, written from the published description of
the block-based two-pass labelling OpenCV uses for 8-connectivity (Grana et al., "Optimized Block-based Connected
Components Labeling with Decision Trees", IEEE TIP 2010 -- BBDT, OpenCV's LabelingGrana; Bolelli et al., "Spaghetti
Labeling", IEEE TIP 2019 -- OpenCV >= 4.5.2), not from orc_contour.h:
  * first scan over 2x2 blocks in raster order; a block with a foreground pixel takes the label(s) of the already
    scanned neighbour blocks (left, upper-left, upper, upper-right) it is 8-connected to THROUGH ACTUAL PIXELS, merging them
    in a union-find whose root is the SMALLER label; with no such neighbour it gets a NEW provisional label (1, 2, ... in
    scan order);
  * flatten: provisional labels in increasing order, a root gets the next final label, a non-root its root's;
  * second scan: every foreground pixel of a block gets the final label of the block.
(The decision trees of BBDT / Spaghetti only decide which neighbours have to be looked at; the equivalence classes and the
scan order of label creation are those of the plain procedure above.)  Two restatements agreeing is weaker than a pin on
OpenCV itself -- that needs the KITTI files (tests/test_gpu_kitti_pin.py) -- but it catches a misread rule."""
import numpy as np
import pytest


def two_pass_block_labels(img):
    """8-connected labelling, numbering as produced by the block-based two-pass algorithm (see module docstring)."""
    rows, cols = img.shape
    nbr, nbc = (rows + 1) // 2, (cols + 1) // 2
    fg = img != 0

    def px(r, c):
        return 0 <= r < rows and 0 <= c < cols and fg[r, c]

    P = [0]                       # union-find over provisional labels, P[i] <= i; index 0 = background
    blab = np.zeros((nbr, nbc), np.int64)

    def find(i):
        while P[i] < i:
            i = P[i]
        return i

    def union(i, j):
        ri, rj = find(i), find(j)
        r = min(ri, rj)
        for k in (i, j):          # path compression towards the smaller root (set_union + set_root of the OpenCV helper)
            while P[k] > r:
                k, P[k] = P[k], r
        P[ri] = P[rj] = r
        return r

    for br in range(nbr):
        for bc in range(nbc):
            r, c = 2 * br, 2 * bc
            a, b, cc_, d = px(r, c), px(r, c + 1), px(r + 1, c), px(r + 1, c + 1)  # a b / c d
            if not (a or b or cc_ or d):
                continue
            lab = 0
            # neighbour blocks already scanned and the pixel adjacencies that connect them to this block
            nbs = []
            if bc > 0 and blab[br, bc - 1]:            # left block: its right column touches our left column
                if (a or cc_) and (px(r, c - 1) or px(r + 1, c - 1)):
                    nbs.append(blab[br, bc - 1])
            if br > 0 and bc > 0 and blab[br - 1, bc - 1]:  # upper-left block: only its lower-right pixel touches our a
                if a and px(r - 1, c - 1):
                    nbs.append(blab[br - 1, bc - 1])
            if br > 0 and blab[br - 1, bc]:            # upper block: its bottom row touches our top row
                if (a and (px(r - 1, c) or px(r - 1, c + 1))) or (b and (px(r - 1, c) or px(r - 1, c + 1))):
                    nbs.append(blab[br - 1, bc])
            if br > 0 and bc + 1 < nbc and blab[br - 1, bc + 1]:  # upper-right block: its lower-left pixel touches our b
                if b and px(r - 1, c + 2):
                    nbs.append(blab[br - 1, bc + 1])
            if not nbs:
                P.append(len(P))
                lab = len(P) - 1
            else:
                lab = nbs[0]
                for o in nbs[1:]:
                    lab = union(lab, o)
                lab = find(lab)
            blab[br, bc] = lab
    # flatten
    k = 1
    for i in range(1, len(P)):
        if P[i] < i:
            P[i] = P[P[i]]
        else:
            P[i] = k
            k += 1
    out = np.zeros((rows, cols), np.int32)
    for br in range(nbr):
        for bc in range(nbc):
            if blab[br, bc]:
                f = P[blab[br, bc]]
                for r in (2 * br, 2 * br + 1):
                    for c in (2 * bc, 2 * bc + 1):
                        if px(r, c):
                            out[r, c] = f
    return out, k


@pytest.mark.parametrize("seed", range(6))
def test_second_restatement_agrees_with_the_oracle_numbering(oracle, seed):
    rng = np.random.default_rng(100 + seed)
    n_checked = n_not_pixel_order = 0
    from scipy import ndimage
    for trial in range(60):
        rows, cols = int(rng.integers(1, 41)), int(rng.integers(1, 41))     # odd and even sizes: partial last blocks
        dens = rng.choice([0.15, 0.3, 0.45, 0.6, 0.8])
        img = (rng.random((rows, cols)) < dens).astype(np.uint8) * 255
        if trial % 5 == 0:   # blobs: the shapes contours have
            yy, xx = np.mgrid[0:rows, 0:cols]
            img[:] = 0
            for _ in range(int(rng.integers(1, 9))):
                cy, cx, ry, rx = rng.uniform(0, rows), rng.uniform(0, cols), rng.uniform(0.6, 6), rng.uniform(0.6, 6)
                img[((yy - cy) / ry) ** 2 + ((xx - cx) / rx) ** 2 <= 1.0] = 255
        a, n_a = two_pass_block_labels(img)
        b, st = oracle.ccl8(img)
        assert n_a == len(st), (rows, cols, n_a, len(st))
        assert np.array_equal(a, b), "numbering differs on a %d x %d patch (seed %d, trial %d)" % (rows, cols, seed, trial)
        n_checked += int(n_a > 2)
        pix, _ = ndimage.label(img, structure=np.ones((3, 3)))   # numbered by first PIXEL in raster order
        n_not_pixel_order += int(not np.array_equal(pix, a))
    assert n_checked > 30   # most patches had several components, so the ORDER was what got compared
    assert n_not_pixel_order > 5   # ... and the block order is a different order than the plain pixel-raster one


def test_the_same_scene_through_shifted_roi_origins(oracle):
    """The numbering is relative to the patch origin: shifting the ROI by one pixel regroups the 2x2 blocks.  Both
    restatements must follow it together (the reference crops every child ROI to its parent's bounding box,
    contour_mng.cpp:309-312, so odd origins are the normal case), and the numbers do move with the origin."""
    rng = np.random.default_rng(7)
    big = (rng.random((48, 48)) < 0.35).astype(np.uint8) * 255
    base, _ = two_pass_block_labels(np.ascontiguousarray(big[0:33, 0:31]))
    moved = 0
    for (r0, c0) in [(0, 0), (1, 0), (0, 1), (1, 1), (3, 6), (7, 2)]:
        patch = np.ascontiguousarray(big[r0:r0 + 33, c0:c0 + 31])
        a, _ = two_pass_block_labels(patch)
        b, _ = oracle.ccl8(patch)
        assert np.array_equal(a, b), (r0, c0)
        if (r0, c0) in ((1, 0), (0, 1), (1, 1)):   # overlap with the unshifted patch: same pixels, possibly other numbers
            ov_a = a[:33 - r0, :31 - c0]
            ov_base = base[r0:, c0:]
            both = (ov_a > 0) & (ov_base > 0)
            moved += int(not np.array_equal(ov_a[both], ov_base[both]))
    assert moved >= 1
