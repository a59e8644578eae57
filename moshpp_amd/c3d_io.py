"""Minimal C3D reader / writer (pure NumPy) for the Stage-II ingest path.

The reference reads captures with ezc3d (src/moshpp/tools/mocap_interface.py:119-127), which is not
available here.  This module parses the public C3D layout directly (header block, parameter section,
float or scaled-int16 point frames; Intel and MIPS byte orders; DEC floats are converted) and returns what
`read_mocap` takes from ezc3d: points[frames, n_points, 3] with invalid samples (residual < 0) as NaN,
POINT:LABELS (+LABELS2..), POINT:RATE.
"""
from __future__ import annotations

import struct

import numpy as np

PROC_INTEL, PROC_DEC, PROC_MIPS = 84, 85, 86


def _dec_to_ieee(raw_u32):
    """DEC VAX F-float stored as two swapped 16-bit words -> IEEE float32."""
    w = raw_u32.astype(np.uint32)
    swapped = ((w & 0xFFFF) << 16) | (w >> 16)
    out = swapped.view(np.float32) / 4.0
    out[swapped == 0] = 0.0
    return out


class _Reader:
    def __init__(self, buf, proc):
        self.buf = buf
        self.proc = proc
        self.end = '>' if proc == PROC_MIPS else '<'

    def u8(self, off): return self.buf[off]
    def i8(self, off): return struct.unpack_from('b', self.buf, off)[0]
    def u16(self, off): return struct.unpack_from(self.end + 'H', self.buf, off)[0]
    def i16(self, off): return struct.unpack_from(self.end + 'h', self.buf, off)[0]

    def f32(self, off):
        if self.proc == PROC_DEC:
            raw = np.frombuffer(self.buf, dtype='<u4', count=1, offset=off)
            return float(_dec_to_ieee(raw.copy())[0])
        return struct.unpack_from(self.end + 'f', self.buf, off)[0]


def read_c3d(fname):
    """-> dict(points[F,N,3] float64 (NaN = invalid), residuals[F,N], labels[list[str]], frame_rate, units,
    parameters{GROUP: {NAME: value}})."""
    with open(fname, 'rb') as f:
        buf = f.read()
    if len(buf) < 512 or buf[1] != 0x50:
        raise ValueError(f'{fname}: not a C3D file')
    param_block = buf[0]
    poff = (param_block - 1) * 512
    proc = buf[poff + 3]
    if proc not in (PROC_INTEL, PROC_DEC, PROC_MIPS):
        raise ValueError(f'{fname}: unknown processor type {proc}')
    rd = _Reader(buf, proc)
    n_points = rd.u16(2)
    n_analog_per_frame = rd.u16(4)
    first_frame, last_frame = rd.u16(6), rd.u16(8)
    scale = rd.f32(12)
    data_block = rd.u16(16)
    analog_per_frame_rate = rd.u16(18)
    frame_rate = rd.f32(20)

    # ---- parameter section
    groups, params = {}, {}
    off = poff + 4
    end_params = poff + 512 * max(int(buf[poff + 2]), 1)
    while off < min(end_params, len(buf) - 2):
        nlen = rd.i8(off)
        gid = rd.i8(off + 1)
        if nlen == 0:
            break
        n = abs(nlen)
        name = buf[off + 2:off + 2 + n].decode('latin-1').strip().upper()
        nxt_off = off + 2 + n
        nxt = rd.i16(nxt_off)
        if gid < 0:
            groups[-gid] = name
            params.setdefault(name, {})
        else:
            p = nxt_off + 2
            dtype = rd.i8(p)
            ndim = rd.u8(p + 1)
            dims = [rd.u8(p + 2 + i) for i in range(ndim)]
            dstart = p + 2 + ndim
            count = int(np.prod(dims)) if ndim else 1
            if dtype == -1:
                raw = buf[dstart:dstart + count]
                if ndim >= 2:
                    width = dims[0]
                    val = [raw[i * width:(i + 1) * width].decode('latin-1').rstrip(' \x00') for i in range(count // max(width, 1))]
                else:
                    val = raw.decode('latin-1').rstrip(' \x00')
            elif dtype == 1:
                val = np.frombuffer(buf, dtype=np.uint8, count=count, offset=dstart).copy()
            elif dtype == 2:
                val = np.frombuffer(buf, dtype=rd.end + 'i2', count=count, offset=dstart).astype(np.int64)
            elif dtype == 4:
                if proc == PROC_DEC:
                    val = _dec_to_ieee(np.frombuffer(buf, dtype='<u4', count=count, offset=dstart).copy()).astype(np.float64)
                else:
                    val = np.frombuffer(buf, dtype=rd.end + 'f4', count=count, offset=dstart).astype(np.float64)
            else:
                val = None
            if isinstance(val, np.ndarray) and ndim == 0:
                val = val.reshape(())
            params.setdefault(gid, {})[name] = val
        if nxt <= 0:
            break
        off = nxt_off + nxt
    named = {}
    for gid, gname in groups.items():
        named[gname] = params.pop(gid, {})
    for k, v in params.items():
        if isinstance(k, str):
            named.setdefault(k, {}).update(v)

    point = named.get('POINT', {})

    def scalar(v, default):
        if v is None:
            return default
        a = np.asarray(v).ravel()
        return a[0] if a.size else default

    n_points = int(scalar(point.get('USED'), n_points))
    if 'SCALE' in point:
        scale = float(scalar(point['SCALE'], scale))
    if 'RATE' in point:
        frame_rate = float(scalar(point['RATE'], frame_rate))
    if 'DATA_START' in point:
        data_block = int(scalar(point['DATA_START'], data_block)) & 0xFFFF
    n_frames = last_frame - first_frame + 1
    if 'FRAMES' in point:
        nf = int(scalar(point['FRAMES'], n_frames)) & 0xFFFF
        if nf > 0 and last_frame == 65535:
            n_frames = nf
    trial = named.get('TRIAL', {})
    if 'ACTUAL_START_FIELD' in trial and 'ACTUAL_END_FIELD' in trial:
        s = np.asarray(trial['ACTUAL_START_FIELD']).ravel().astype(np.int64)
        e = np.asarray(trial['ACTUAL_END_FIELD']).ravel().astype(np.int64)
        if s.size >= 2 and e.size >= 2:
            start = (s[0] & 0xFFFF) | ((s[1] & 0xFFFF) << 16)
            endf = (e[0] & 0xFFFF) | ((e[1] & 0xFFFF) << 16)
            if endf >= start and endf - start + 1 > n_frames:
                n_frames = int(endf - start + 1)
    labels = []
    for key in ['LABELS'] + [f'LABELS{i}' for i in range(2, 20)]:
        v = point.get(key)
        if v is None:
            break
        labels += list(v) if isinstance(v, list) else [v]
    labels = [l.strip() for l in labels][:n_points] if labels else []

    # ---- data section
    doff = (data_block - 1) * 512
    is_float = scale < 0
    words_per_frame = 4 * n_points + n_analog_per_frame
    wsize = 4 if is_float else 2
    avail = (len(buf) - doff) // (words_per_frame * wsize) if words_per_frame else 0
    n_frames = max(0, min(n_frames, avail))
    if is_float:
        if proc == PROC_DEC:
            raw = _dec_to_ieee(np.frombuffer(buf, dtype='<u4', count=n_frames * words_per_frame, offset=doff).copy())
        else:
            raw = np.frombuffer(buf, dtype=rd.end + 'f4', count=n_frames * words_per_frame, offset=doff)
        frames = raw.reshape(n_frames, words_per_frame)[:, :4 * n_points].reshape(n_frames, n_points, 4).astype(np.float64)
        xyz = frames[:, :, :3].copy()
        fourth = frames[:, :, 3]
        invalid = fourth < 0
        residual = np.where(invalid, -1.0, (fourth.astype(np.int64) & 0xFF) * abs(scale))
    else:
        raw = np.frombuffer(buf, dtype=rd.end + 'i2', count=n_frames * words_per_frame, offset=doff)
        frames = raw.reshape(n_frames, words_per_frame)[:, :4 * n_points].reshape(n_frames, n_points, 4)
        xyz = frames[:, :, :3].astype(np.float64) * scale
        fourth = frames[:, :, 3]
        invalid = fourth < 0
        residual = np.where(invalid, -1.0, (fourth.astype(np.int64) & 0xFF) * abs(scale))
    xyz[invalid] = np.nan
    units = point.get('UNITS', 'mm')
    if isinstance(units, list):
        units = units[0] if units else 'mm'
    return dict(points=xyz, residuals=residual, labels=labels, frame_rate=frame_rate, units=units,
                parameters=named, first_frame=first_frame, analog_per_frame=analog_per_frame_rate)


# --------------------------------------------------------------------------------------------------
def _ieee_to_dec_bytes(vals):
    """float32 values -> DEC VAX F-float bytes (exponent bias +2 = x4, the two 16-bit words swapped)."""
    w = (np.asarray(vals, dtype='<f4') * np.float32(4.0)).astype('<f4').view('<u4')
    w = np.where(np.asarray(vals, dtype='<f4') == 0, np.uint32(0), w)
    return (((w & 0xFFFF) << 16) | (w >> 16)).astype('<u4').tobytes()


class _Enc:
    """Byte-level encoders of one processor format (Intel little-endian IEEE, MIPS big-endian IEEE, DEC)."""

    def __init__(self, proc):
        self.proc = proc
        self.e = '>' if proc == PROC_MIPS else '<'

    def i16(self, v): return struct.pack(self.e + 'h', int(v))
    def u16(self, v): return struct.pack(self.e + 'H', int(v))

    def f32(self, v):
        return _ieee_to_dec_bytes([v]) if self.proc == PROC_DEC else struct.pack(self.e + 'f', float(v))

    def f32_array(self, a):
        return _ieee_to_dec_bytes(a) if self.proc == PROC_DEC else np.asarray(a).astype(self.e + 'f4').tobytes()

    def i16_array(self, a):
        return np.asarray(a).astype(self.e + 'i2').tobytes()


def _param(name, gid, dtype, dims, data_bytes, desc=b'', enc=None):
    enc = enc or _Enc(PROC_INTEL)
    body = struct.pack('<b', dtype) + struct.pack('<B', len(dims)) + bytes(dims) + data_bytes
    body += struct.pack('<B', len(desc)) + desc
    nm = name.encode('latin-1')
    return struct.pack('<bb', len(nm), gid) + nm + enc.i16(len(body) + 2) + body


def _group(name, gid, desc=b'', enc=None):
    enc = enc or _Enc(PROC_INTEL)
    nm = name.encode('latin-1')
    body = struct.pack('<B', len(desc)) + desc
    return struct.pack('<bb', len(nm), -gid) + nm + enc.i16(len(body) + 2) + body


def write_c3d(fname, points, labels, frame_rate=120.0, units='mm', processor=PROC_INTEL, int_scale=None):
    """C3D writer.  points[F,N,3] in file units; NaN or all-zero samples are written as invalid (residual -1), like the
    reference's writer (mocap_interface.py:51-84).  Default: Intel / float format (what the reference writes);
    `processor` PROC_MIPS / PROC_DEC and `int_scale` (> 0: scaled-int16 point format, POINT:SCALE = int_scale) produce the
    other on-disk variants the format allows (reference reader: tools/c3d.py:35-60, 1293-1385)."""
    points = np.asarray(points, dtype=np.float64)
    F, N, _ = points.shape
    assert len(labels) == N
    if F > 65535:
        raise ValueError('this writer stores at most 65535 frames')
    enc = _Enc(processor)
    invalid = np.logical_or(np.isnan(points).any(-1), (points == 0).all(-1))
    scale = -1.0 if int_scale is None else float(int_scale)
    if int_scale is None:
        data = np.zeros((F, N, 4), dtype=np.float32)
        data[:, :, :3] = np.where(invalid[..., None], 0.0, points)
        data[:, :, 3] = np.where(invalid, -1.0, 0.0)
        raw = enc.f32_array(data.ravel())
    else:
        q = np.rint(np.where(invalid[..., None], 0.0, points) / scale)
        if np.abs(q).max() > 32767:
            raise ValueError('int_scale too small for the coordinate range')
        data = np.zeros((F, N, 4), dtype=np.int16)
        data[:, :, :3] = q.astype(np.int16)
        data[:, :, 3] = np.where(invalid, -1, 0)   # valid: residual byte 0, camera mask 0
        raw = enc.i16_array(data.ravel())
    width = max(4, max((len(l) for l in labels), default=4))
    P = lambda *a, **k: _param(*a, enc=enc, **k)   # noqa: E731
    recs = _group('POINT', 1, enc=enc) + _group('ANALOG', 2, enc=enc)
    recs += P('USED', 1, 2, [], enc.i16(N))
    recs += P('FRAMES', 1, 2, [], enc.u16(F))
    recs += P('SCALE', 1, 4, [], enc.f32(scale))
    recs += P('RATE', 1, 4, [], enc.f32(float(frame_rate)))
    recs += P('UNITS', 1, -1, [len(units)], units.encode('latin-1'))
    for blk in range(0, max(N, 1), 255):
        chunk = labels[blk:blk + 255]
        key = 'LABELS' if blk == 0 else f'LABELS{blk // 255 + 1}'
        rawl = b''.join(l.encode('latin-1').ljust(width)[:width] for l in chunk)
        recs += P(key, 1, -1, [width, len(chunk)], rawl)
    recs += P('USED', 2, 2, [], enc.i16(0))
    # DATA_START needs the final parameter-section size: fixed-size record, so compute first
    ds_len = len(P('DATA_START', 1, 2, [], enc.i16(0)))
    n_pblocks = (4 + len(recs) + ds_len + 2 + 511) // 512
    data_start = 2 + n_pblocks
    recs += P('DATA_START', 1, 2, [], enc.i16(data_start))
    psec = bytes([1, 0x50, n_pblocks, processor]) + recs + b'\x00\x00'
    psec = psec.ljust(n_pblocks * 512, b'\x00')
    hdr = bytearray(512)
    hdr[0] = 2
    hdr[1] = 0x50
    hdr[2:12] = enc.u16(N) + enc.u16(0) + enc.u16(1) + enc.u16(F) + enc.u16(0)
    hdr[12:16] = enc.f32(scale)
    hdr[16:20] = enc.u16(data_start) + enc.u16(0)
    hdr[20:24] = enc.f32(float(frame_rate))
    with open(fname, 'wb') as f:
        f.write(bytes(hdr))
        f.write(psec)
        f.write(raw)
        pad = (-len(raw)) % 512
        f.write(b'\x00' * pad)
