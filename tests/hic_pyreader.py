"""Test infrastructure: a second, independent reading of the Juicer `.hic` format -- pure Python (struct + zlib from the standard
library), written from the format description (github.com/aidenlab/hic-format, HiCFormatV8.md / HiCFormatV9.md, and the prose
of straw's README), NOT from mustache_amd/csrc/hic_reader.cpp and NOT from tests/hic_writer.py: it shares no code, helper or
constant with either.  hic-straw itself cannot be had offline, so the native reader's block decoding is pinned on TWO
independent readings instead of one: what tests/hic_writer.py writes must come back identically through this reader and through
libmustache_io.so, and hand-assembled blocks of kinds the writer never emits must decode identically in both
(tests/test_hic_two_readings.py).

Format as read here (little-endian throughout):

  file     := magic "HIC\\0" | version:i32 | footerPosition:i64 | genomeId:cstr | [v9: normVectorIndexPosition:i64,
              normVectorIndexLength:i64] | nAttributes:i32 {key:cstr value:cstr} | nChromosomes:i32 {name:cstr, length: v9 i64 /
              else i32} | nBpRes:i32 {i32} | nFragRes:i32 {i32} | ...
  footer   := nBytes: v9 i64 / else i32 | nEntries:i32 {key:cstr "a_b", position:i64, sizeInBytes:i32} | expected-value vectors |
              normalised expected-value vectors | normalisation-vector index
  expected := n:i32 { [normalised: type:cstr] unit:cstr binSize:i32 nValues: v9 i64 / else i32  values: v9 f32 / else f64
              nChrScale:i32 {chrIndex:i32 factor: v9 f32 / else f64} }
  nvindex  := n:i32 { type:cstr chrIdx:i32 unit:cstr binSize:i32 position:i64 sizeInBytes: v9 i64 / else i32 }
  normvec  := nValues: v9 i64 / else i32 | values: v9 f32 / else f64
  matrix   := chr1Idx:i32 chr2Idx:i32 nResolutions:i32 { unit:cstr zoomIndex:i32 sumCounts:f32 occupied:f32 stdDev:f32
              percent95:f32 binSize:i32 blockBinCount:i32 blockColumnCount:i32 blockCount:i32 {blockNumber:i32
              blockPosition:i64 blockSizeBytes:i32} }
  block    := zlib( nRecords:i32 ... )
              v6     : nRecords x {binX:i32 binY:i32 value:f32}
              v7, v8 : binXOffset:i32 binYOffset:i32 useShort:u8 (0 = counts are i16, else f32) matrixType:u8
              v9     : binXOffset:i32 binYOffset:i32 useShort:u8 useShortBinX:u8 useShortBinY:u8 (0 = i16, else i32) matrixType:u8
              type 1 : rowCount (i16, v9 long-Y: i32) { rowNumber (same width) recordCount (i16, v9 long-X: i32)
                       { binColumn (same width as recordCount) value (i16 | f32) } };  bin = offset + relative number
              type 2 : nPoints:i32 width:i16 nPoints x value (i16 with -32768 = no data | f32 with NaN = no data), point i sits
                       at row i // width, column i % width
  The value of a record under normalisation N is counts / (N[binX] * N[binY]).
"""
import math
import struct
import zlib


class _Bytes:
    def __init__(self, data, at=0):
        self.data, self.at = data, at

    def take(self, fmt):
        size = struct.calcsize("<" + fmt)
        vals = struct.unpack_from("<" + fmt, self.data, self.at)
        self.at += size
        return vals[0] if len(vals) == 1 else vals

    def text(self):
        end = self.data.index(b"\x00", self.at)
        out = self.data[self.at:end].decode()
        self.at = end + 1
        return out


class PyHic:
    """Whole file in memory; `records(chromosome, resolution, norm)` -> list of (binX, binY, value) for the intra matrix."""

    def __init__(self, path):
        with open(path, "rb") as fh:
            self.data = fh.read()
        r = _Bytes(self.data)
        if r.text() != "HIC":
            raise ValueError("not a .hic file")
        self.version = r.take("i")
        self.footer_at = r.take("q")
        self.genome = r.text()
        self.nvi = None
        new = self.version >= 9
        if new:
            self.nvi = r.take("qq")
        for _ in range(r.take("i")):
            r.text(), r.text()
        self.chromosomes = []
        for _ in range(r.take("i")):
            name = r.text()
            self.chromosomes.append((name, r.take("q") if new else r.take("i")))
        self.resolutions = [r.take("i") for _ in range(r.take("i"))]
        # footer: matrix directory
        f = _Bytes(self.data, self.footer_at)
        f.take("q") if new else f.take("i")
        self.directory = {}
        for _ in range(f.take("i")):
            key = f.text()
            self.directory[key] = f.take("qi")
        # expected values (skipped), then the normalisation-vector index
        for normalised in (False, True):
            for _ in range(f.take("i")):
                if normalised:
                    f.text()
                f.text()
                f.take("i")
                count = f.take("q") if new else f.take("i")
                f.at += count * (4 if new else 8)
                factors = f.take("i")
                f.at += factors * (4 + (4 if new else 8))
        if new and self.nvi and self.nvi[0] > 0:
            f = _Bytes(self.data, self.nvi[0])
        self.norm_index = {}
        for _ in range(f.take("i")):
            typ, chrom, unit, binsize = f.text(), f.take("i"), f.text(), f.take("i")
            where = f.take("q")
            size = f.take("q") if new else f.take("i")
            self.norm_index[(typ, chrom, unit, binsize)] = (where, size)

    def chromosome_index(self, name):
        for i, (n, _) in enumerate(self.chromosomes):
            if n == name:
                return i
        raise KeyError(name)

    def norm_vector(self, typ, chrom_index, resolution):
        where, _ = self.norm_index[(typ, chrom_index, "BP", resolution)]
        r = _Bytes(self.data, where)
        if self.version >= 9:
            n = r.take("q")
            return list(struct.unpack_from("<%df" % n, self.data, r.at))
        n = r.take("i")
        return list(struct.unpack_from("<%dd" % n, self.data, r.at))

    def zoom(self, chrom_index, resolution):
        where, _ = self.directory["%d_%d" % (chrom_index, chrom_index)]
        r = _Bytes(self.data, where)
        r.take("ii")
        for _ in range(r.take("i")):
            unit = r.text()
            r.take("i")
            r.take("ffff")
            binsize, per_block, columns, nblocks = r.take("iiii")
            entries = [r.take("iqi") for _ in range(nblocks)]
            if unit == "BP" and binsize == resolution:
                return per_block, columns, entries
        raise KeyError(resolution)

    def block_records(self, position, size):
        """Raw (binX, binY, counts) of one block, file order."""
        body = zlib.decompress(self.data[position:position + size])
        r = _Bytes(body)
        n_records = r.take("i")
        out = []
        if self.version < 7:
            for _ in range(n_records):
                out.append(r.take("iif"))
            return out
        x0, y0 = r.take("ii")
        counts_are_short = r.take("B") == 0
        x_short = y_short = True
        if self.version >= 9:
            x_short = r.take("B") == 0
            y_short = r.take("B") == 0
        kind = r.take("B")
        cfmt = "h" if counts_are_short else "f"
        if kind == 1:
            xfmt, yfmt = ("h" if x_short else "i"), ("h" if y_short else "i")
            for _ in range(r.take(yfmt)):
                row = r.take(yfmt)
                for _ in range(r.take(xfmt)):
                    col = r.take(xfmt)
                    out.append((x0 + col, y0 + row, float(r.take(cfmt))))
        elif kind == 2:
            points = r.take("i")
            width = r.take("h")
            for i in range(points):
                val = r.take(cfmt)
                missing = (val == -32768) if counts_are_short else math.isnan(val)
                if not missing:
                    out.append((x0 + i % width, y0 + i // width, float(val)))
        else:
            raise ValueError("unknown block kind %d" % kind)
        return out

    def records(self, chromosome, resolution, norm="NONE"):
        """Every record of the chromosome's intra matrix: (binX, binY, value as float32) with value = counts / (norm[x] *
        norm[y]) evaluated in double and rounded to float32; NaN values and values <= 0 are kept (the caller filters)."""
        ci = self.chromosome_index(chromosome)
        _, _, entries = self.zoom(ci, resolution)
        vec = None if norm in (None, "", "NONE") else self.norm_vector(norm, ci, resolution)
        out = []
        for _, where, size in entries:
            if size <= 0:
                continue
            for bx, by, c in self.block_records(where, size):
                if vec is not None:
                    if not (0 <= bx < len(vec) and 0 <= by < len(vec)):
                        continue                                  # outside the vector: no value defined
                    denom = vec[bx] * vec[by]
                    c = c / denom if denom != 0 else (math.nan if c == 0 else math.copysign(math.inf, c) * math.copysign(1.0, denom))
                out.append((bx, by, struct.unpack("<f", struct.pack("<f", c))[0] if math.isfinite(c) and abs(c) < 3.4e38 else c))
        return out
