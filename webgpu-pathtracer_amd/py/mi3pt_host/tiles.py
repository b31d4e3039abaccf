"""Image tile split across ranks: rows are dealt to ranks in blocks of
`block_rows`, round after round, back and forth (rank r owns block r of even rounds and block nranks - 1 - r of odd ones: a cost
that rises or falls down the image is shared out evenly); every rank keeps a compact [local_rows, width, 4] image.  The
seed of a pixel depends only on its GLOBAL index and the frame (raytrace.wgsl:435-436),
so any split renders identical pixels; one gather of the HDR accumulation buffers plus
this de-interleave reassembles the picture."""
import numpy as np


def local_rows_of(height, rank, nranks, block_rows):
    """Global row indices held by `rank`, in local order -- asked of the LIBRARY (mi3pt_tile_local_rows / mi3pt_tile_global_row, ABI 3: hosts
    must not keep a copy of the deal's formula; round-5 advice: this function used to)."""
    from . import capi
    n = capi.tile_local_rows(height, rank, nranks, block_rows)
    return [capi.tile_global_row(ly, rank, nranks, block_rows) for ly in range(n)]


def deinterleave_rows(parts, height, nranks, block_rows):
    """parts[r]: rank r's (possibly zero-padded) compact image -> the whole image."""
    width = parts[0].shape[1]
    out = np.zeros((height, width, parts[0].shape[2]), parts[0].dtype)
    for r in range(nranks):
        rows = local_rows_of(height, r, nranks, block_rows)
        out[rows] = np.asarray(parts[r])[:len(rows)]
    return out


def balanced_bands(tile_cost, height, nranks, floor_per_pixel=0.0):
    """Contiguous bands of whole 8-row tile rows with (nearly) equal measured cost: boundaries [b0 = 0, b1, ..., bN = height]
    in image rows; rank r renders rows [b_r, b_{r+1}) (mi3pt_set_rows).  tile_cost: what mi3pt_measure_tile_cost returned for
    the WHOLE image, shape (tile rows, tile columns).  floor_per_pixel: a fixed cost per pixel on top of the measured one
    (refill, camera ray: what even a sky pixel costs), in the measure's units.  Pure integer arithmetic -- but the measurement
    itself repeats only to a fraction of a per cent (test counts of the culling walks depend on wave scheduling), so in a
    multi-process job one rank measures and broadcasts these bounds."""
    cost = np.asarray(tile_cost, np.int64)
    row_cost = cost.sum(axis=1) + int(round(floor_per_pixel * 64)) * cost.shape[1]
    nrows_t = len(row_cost)
    if nrows_t <= nranks:           # no more tile rows than ranks: one each, the last ranks none
        return [min(8 * r, height) for r in range(nranks)] + [height]
    total = int(row_cost.sum())
    bounds = [0]
    acc, r = 0, 1
    for t in range(nrows_t):
        acc += int(row_cost[t])
        # cut after tile row t once this rank has its share -- at most ONE cut per tile row (a row that holds several ranks' shares
        # used to be cut several times at the same place: empty bands, idle ranks; round-4 advice) and leaving at least one tile row
        # for every rank still to come (when exactly that many rows are left, every one of them is a cut)
        left, waiting = nrows_t - (t + 1), nranks - r
        if r < nranks and left >= waiting and (acc * nranks >= total * r or left == waiting):
            bounds.append(min((t + 1) * 8, height))
            r += 1
    while len(bounds) < nranks:
        bounds.append(height)
    bounds.append(height)
    return bounds


def stack_bands(parts, bounds):
    """parts[r]: rank r's band image (rows bounds[r] .. bounds[r + 1]) -> the whole image."""
    return np.concatenate([np.asarray(p)[: bounds[r + 1] - bounds[r]] for r, p in enumerate(parts)], axis=0)
