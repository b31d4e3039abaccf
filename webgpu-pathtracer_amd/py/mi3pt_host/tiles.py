"""Image tile split across ranks: rows are dealt to ranks in blocks of
`block_rows`, round robin; every rank keeps a compact [local_rows, width, 4] image.  The
seed of a pixel depends only on its GLOBAL index and the frame (raytrace.wgsl:435-436),
so any split renders identical pixels; one gather of the HDR accumulation buffers plus
this de-interleave reassembles the picture."""
import numpy as np


def local_rows_of(height, rank, nranks, block_rows):
    """Global row indices held by `rank`, in local order."""
    return [y for y in range(height) if (y // block_rows) % nranks == rank]


def deinterleave_rows(parts, height, nranks, block_rows):
    """parts[r]: rank r's (possibly zero-padded) compact image -> the whole image."""
    width = parts[0].shape[1]
    out = np.zeros((height, width, parts[0].shape[2]), parts[0].dtype)
    for r in range(nranks):
        rows = local_rows_of(height, r, nranks, block_rows)
        out[rows] = np.asarray(parts[r])[:len(rows)]
    return out
