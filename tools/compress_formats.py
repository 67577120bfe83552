"""Writers of the compressed FASTQ containers the file path reads, for tests, tools and the bench's tiers (nothing here
is product code, and nothing here touches the oracle):
  bgzf_bytes          BGZF (what bgzip writes): gzip members of at most 64 KiB with the 'BC' extra field;
  gzip_one_member     ONE ordinary gzip member -- a single DEFLATE stream, as `gzip` and `pigz` write -- compressed on
                      several threads the way pigz does it: independent pieces, each closed by a full flush (an empty
                      stored block, byte aligned), concatenated, one CRC-32 / ISIZE trailer."""
import struct
import zlib


def bgzf_bytes(data, block=0xFF00, level=6, threads=0):
    """`data` as a BGZF file: gzip members of at most 64 KiB, each with the 'BC' extra field holding its compressed
    size, closed by the empty end-of-file member.  `threads` > 1: the members are compressed on a thread pool (zlib
    releases the GIL)."""
    def member(i):
        chunk = data[i:i + block]
        co = zlib.compressobj(level, zlib.DEFLATED, -15)
        comp = co.compress(chunk) + co.flush()
        return (b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff\x06\x00BC\x02\x00" + struct.pack("<H", len(comp) + 25) + comp
                + struct.pack("<II", zlib.crc32(chunk) & 0xFFFFFFFF, len(chunk)))
    starts = range(0, len(data), block)
    if threads > 1:
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(threads) as ex:
            out = list(ex.map(member, starts, chunksize=64))
    else:
        out = [member(i) for i in starts]
    out.append(bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000"))
    return b"".join(out)


def gzip_one_member(data, level=1, threads=8, piece=1 << 20):
    """`data` as one gzip member (header, one DEFLATE stream, CRC-32, ISIZE)."""
    view = memoryview(data)
    starts = list(range(0, len(view), piece)) or [0]

    def part(i):
        co = zlib.compressobj(level, zlib.DEFLATED, -15)
        body = co.compress(view[i:i + piece])
        last = i == starts[-1]
        return body + co.flush(zlib.Z_FINISH if last else zlib.Z_FULL_FLUSH), zlib.crc32(view[i:i + piece])

    if threads > 1 and len(starts) > 1:
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(threads) as ex:
            parts = list(ex.map(part, starts))
    else:
        parts = [part(i) for i in starts]
    crc = zlib.crc32(b"")
    for (_, c), i in zip(parts, starts):
        crc = _crc32_combine(crc, c, min(piece, len(view) - i))
    return (b"\x1f\x8b\x08\x00\x00\x00\x00\x00\x00\xff" + b"".join(p for p, _ in parts)
            + struct.pack("<II", crc & 0xFFFFFFFF, len(view) & 0xFFFFFFFF))


def _crc32_combine(crc1, crc2, len2):
    """zlib's crc32_combine (not exported by Python's zlib): CRC-32 of A + B from those of A and B and len(B)."""
    if len2 <= 0:
        return crc1

    def times(mat, vec):
        s, i = 0, 0
        while vec:
            if vec & 1:
                s ^= mat[i]
            vec >>= 1
            i += 1
        return s

    def square(mat):
        return [times(mat, mat[n]) for n in range(32)]
    odd = [0xEDB88320] + [1 << n for n in range(31)]        # the operator for one zero bit
    even = square(odd)                                        # two zero bits
    odd = square(even)                                        # four
    while True:
        even = square(odd)                                    # (first pass: one zero byte)
        if len2 & 1:
            crc1 = times(even, crc1)
        len2 >>= 1
        if not len2:
            break
        odd = square(even)
        if len2 & 1:
            crc1 = times(odd, crc1)
        len2 >>= 1
        if not len2:
            break
    return crc1 ^ crc2
