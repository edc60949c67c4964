"""NumPy restatement of mst_band_scatter_hic_rows (csrc/mst_hic_rows.hip) -- test infrastructure: decodes the RAW slabs of
mst_hic_rawstream_* (include/mustache_io.h, include/mustache_hicrow.h) on the CPU, so that the host half of the raw `.hic` read
can be held to the host decoder without a GPU, and the kernel to this on one.  Same order of tests as hic_reader.cpp's emit()."""
import numpy as np

COUNT_MASK, SHORT_COUNTS, INT_COLUMNS, DENSE = 0x0FFFFFFF, 0x10000000, 0x20000000, 0x40000000


def decode_slab(payload, directory, norm, max_dist, y_limit=0):
    """payload: uint8 array; directory: uint8 array of 16-byte entries {off u32, y i32, x_off i32, count|flags u32}.
    -> (binX int64, binY int64, value float32) of the records the kernel keeps (before its band-size test)."""
    ent = np.frombuffer(np.ascontiguousarray(directory).tobytes(), dtype=np.dtype([("off", "<u4"), ("y", "<i4"), ("xo", "<i4"), ("cf", "<u4")]))
    xs, ys, vs = [], [], []
    for e in ent:
        cnt = int(e["cf"]) & COUNT_MASK
        short_c, int_x, dense = bool(e["cf"] & SHORT_COUNTS), bool(e["cf"] & INT_COLUMNS), bool(e["cf"] & DENSE)
        csz, vsz = (0 if dense else (4 if int_x else 2)), (2 if short_c else 4)
        raw = np.ascontiguousarray(payload[int(e["off"]):int(e["off"]) + cnt * (csz + vsz)]).reshape(cnt, csz + vsz)
        if dense:
            col = np.arange(cnt, dtype=np.int64)
        else:
            col = np.ascontiguousarray(raw[:, :csz]).view("<i4" if int_x else "<i2").reshape(-1).astype(np.int64)
        cb = np.ascontiguousarray(raw[:, csz:])
        if short_c:
            s = cb.view("<i2").reshape(-1)
            ok = (s != -32768) if dense else np.ones(cnt, bool)
            val = s.astype(np.float32)
        else:
            val = cb.view("<f4").reshape(-1)
            ok = ~np.isnan(val) if dense else np.ones(cnt, bool)
        bx, by = int(e["xo"]) + col, np.full(cnt, int(e["y"]), np.int64)
        lo, hi = np.minimum(bx, by), np.maximum(bx, by)
        if max_dist >= 0:
            ok &= hi - lo <= max_dist
        c = val.copy()
        if norm is not None:
            inside = (lo >= 0) & (hi < len(norm))
            ok &= inside
            with np.errstate(all="ignore"):
                c = (val.astype(np.float64) / (norm[np.where(inside, lo, 0)] * norm[np.where(inside, hi, 0)])).astype(np.float32)
        with np.errstate(invalid="ignore"):
            ok &= ~np.isnan(c) & (c > 0)
        if y_limit > 0:
            ok &= hi < y_limit
        xs.append(lo[ok]); ys.append(hi[ok]); vs.append(c[ok])
    cat = lambda parts, dt: np.concatenate(parts) if parts else np.zeros(0, dt)
    return cat(xs, np.int64), cat(ys, np.int64), cat(vs, np.float32)
