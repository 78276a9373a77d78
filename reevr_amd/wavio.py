"""Minimal RIFF/WAVE reader and writer for the offline renderer: PCM 16/24/32-bit
and IEEE float32, any channel count. Returns / takes float32 arrays shaped (channels, frames)."""
from __future__ import annotations

import struct

import numpy as np


def read_wav(path: str):
    with open(path, "rb") as f:
        data = f.read()
    if data[:4] != b"RIFF" or data[8:12] != b"WAVE":
        raise ValueError(f"{path}: not a RIFF/WAVE file")
    pos, fmt, pcm = 12, None, None
    while pos + 8 <= len(data):
        cid, size = data[pos:pos + 4], struct.unpack_from("<I", data, pos + 4)[0]
        body = data[pos + 8:pos + 8 + size]
        if cid == b"fmt ":
            tag, nch, sr, _, _, bits = struct.unpack_from("<HHIIHH", body, 0)
            if tag == 0xFFFE and len(body) >= 26:          # WAVE_FORMAT_EXTENSIBLE: real tag in the sub-format GUID
                tag = struct.unpack_from("<H", body, 24)[0]
            fmt = (tag, nch, sr, bits)
        elif cid == b"data":
            pcm = body
        pos += 8 + size + (size & 1)
    if fmt is None or pcm is None:
        raise ValueError(f"{path}: missing fmt or data chunk")
    tag, nch, sr, bits = fmt
    if tag == 3 and bits == 32:
        x = np.frombuffer(pcm, "<f4").astype(np.float32)
    elif tag == 1 and bits == 16:
        x = np.frombuffer(pcm, "<i2").astype(np.float32) / np.float32(32768.0)
    elif tag == 1 and bits == 32:
        x = (np.frombuffer(pcm, "<i4").astype(np.float64) / 2147483648.0).astype(np.float32)
    elif tag == 1 and bits == 24:
        b = np.frombuffer(pcm[:len(pcm) // 3 * 3], np.uint8).reshape(-1, 3).astype(np.int32)
        v = b[:, 0] | (b[:, 1] << 8) | (b[:, 2] << 16)
        v = np.where(v & 0x800000, v - 0x1000000, v)
        x = (v.astype(np.float64) / 8388608.0).astype(np.float32)
    else:
        raise ValueError(f"{path}: unsupported format tag {tag} / {bits} bits")
    frames = x.size // nch
    return np.ascontiguousarray(x[:frames * nch].reshape(frames, nch).T), int(sr)


def write_wav(path: str, x: np.ndarray, sr: int, float32: bool = True):
    x = np.atleast_2d(np.asarray(x, np.float32))
    nch, frames = x.shape
    inter = np.ascontiguousarray(x.T)
    if float32:
        tag, bits, payload = 3, 32, inter.astype("<f4").tobytes()
    else:
        tag, bits = 1, 16
        payload = np.clip(np.round(inter * 32767.0), -32768, 32767).astype("<i2").tobytes()
    block = nch * bits // 8
    fmt = struct.pack("<HHIIHH", tag, nch, sr, sr * block, block, bits)
    with open(path, "wb") as f:
        f.write(b"RIFF" + struct.pack("<I", 4 + 8 + len(fmt) + 8 + len(payload)) + b"WAVE")
        f.write(b"fmt " + struct.pack("<I", len(fmt)) + fmt)
        f.write(b"data" + struct.pack("<I", len(payload)) + payload)
