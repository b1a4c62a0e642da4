"""WAV / .sac container layer around the GPU frame encoder.

Format, from the reference (slmdev/sac v0.7.25):

* `.sac` file = header | 16-byte MD5 of the WAV sample bytes | frame records back to back
  (`Sac::WriteSACHeader`, `/root/reference/src/file/sac.cpp:15-38`; `Codec::EncodeFile`,
  `libsac/libsac.cpp:782-855`).  Header: "SAC2", u16 channels, u32 sample rate, u16 bits per
  sample, u32 samples per channel, u8 max frame length in seconds, u8 0, u32 metadata size,
  metadata.  All little endian.
* metadata = the WAV's chunks so that the decoder can rebuild the file byte for byte
  (`Chunks::PackMetaData / UnpackMetaData`, `file/wav.cpp:24-53`): for every chunk u32 id, u32 size,
  then the payload -- 4 bytes ("WAVE") for RIFF, nothing for `data`, the word-aligned body for
  everything else.  Chunks are recorded the way `Wav::ReadHeader` walks the file
  (`file/wav.cpp:166-263`): parsing stops after a `data` chunk that reaches the end of the file.
* a frame record is what `FrameCoder::WriteEncoded` writes (`libsac.cpp:565-578`), produced here by
  `Context.encode_frames`.

Host-side file code only; 8- and 16-bit PCM (the scope of the GPU path).
"""
from __future__ import annotations

import hashlib
import struct
from dataclasses import dataclass, field

import numpy as np

ID_RIFF, ID_FMT, ID_DATA = 0x46464952, 0x20746D66, 0x61746164


def _align(n: int) -> int:
    return n + (n & 1)


@dataclass
class WavInfo:
    numchannels: int = 0
    samplerate: int = 0
    bitspersample: int = 0
    blockalign: int = 0
    numsamples: int = 0
    chunks: list = field(default_factory=list)     # (id, size, payload bytes) in file order
    data: bytes = b""                              # the sample bytes (numsamples * blockalign)

    @property
    def metadatasize(self) -> int:
        return sum(8 + len(p) for _, _, p in self.chunks)


def parse_wav(raw: bytes) -> WavInfo:
    """Wav::ReadHeader (file/wav.cpp:166-263)."""
    w = WavInfo()
    if len(raw) < 12 or raw[:4] != b"RIFF" or raw[8:12] != b"WAVE":
        raise ValueError("not a RIFF/WAVE file")
    w.chunks.append((ID_RIFF, struct.unpack_from("<I", raw, 4)[0], raw[8:12]))
    pos, size = 12, len(raw)
    have_data = False
    while pos + 8 <= size:
        cid, csz = struct.unpack_from("<II", raw, pos)
        pos += 8
        if cid == ID_FMT:
            if csz not in (16, 18, 40):
                raise ValueError(f"invalid fmt chunk size {csz}")
            body = raw[pos: pos + csz]
            w.chunks.append((cid, csz, body))
            fmt, w.numchannels, w.samplerate, _, w.blockalign, w.bitspersample = struct.unpack_from("<HHIIHH", body, 0)
            if csz >= 18 and struct.unpack_from("<H", body, 16)[0] >= 22:
                w.bitspersample = struct.unpack_from("<H", body, 18)[0]
                fmt = struct.unpack_from("<H", body, 24)[0]
            if fmt != 1:
                raise ValueError("only PCM is supported")
            pos += csz
        elif cid == ID_DATA:
            w.chunks.append((cid, csz, b""))
            have_data = True
            end = pos + _align(csz)
            nbytes = csz
            if end >= size:                       # last chunk: stop (truncated files keep what is there)
                if end > size:
                    nbytes = (size - pos) // w.blockalign * w.blockalign
                w.numsamples = nbytes // w.blockalign
                w.data = raw[pos: pos + w.numsamples * w.blockalign]
                break
            w.numsamples = csz // w.blockalign
            w.data = raw[pos: pos + w.numsamples * w.blockalign]
            pos += csz                             # (the reference does not word-align this seek)
        else:
            n = _align(csz)
            w.chunks.append((cid, csz, raw[pos: pos + n]))
            pos += n
        if pos == size:
            break
    if not have_data or w.blockalign == 0:
        raise ValueError("no fmt/data chunk")
    return w


def pack_metadata(chunks) -> bytes:
    """Chunks::PackMetaData (file/wav.cpp:24-36)."""
    return b"".join(struct.pack("<II", cid, csz) + payload for cid, csz, payload in chunks)


def unpack_metadata(meta: bytes):
    """Chunks::UnpackMetaData (file/wav.cpp:38-53)."""
    chunks, ofs = [], 0
    while ofs < len(meta):
        cid, csz = struct.unpack_from("<II", meta, ofs)
        ofs += 8
        n = 4 if cid == ID_RIFF else (0 if cid == ID_DATA else _align(csz))
        chunks.append((cid, csz, meta[ofs: ofs + n]))
        ofs += n
    return chunks


def pcm_from_wav(w: WavInfo) -> np.ndarray:
    """Wav::ReadSamples sample unpacking (file/wav.cpp:91-108) -> int32 [nch, n]."""
    csize = w.blockalign // w.numchannels
    if csize == 1:
        a = np.frombuffer(w.data, np.uint8).astype(np.int32) - 128
    elif csize == 2:
        a = np.frombuffer(w.data, "<i2").astype(np.int32)
    elif csize == 3:                                    # wav.cpp:109-121: three little-endian bytes, sign from the top one
        b = np.frombuffer(w.data, np.uint8).reshape(-1, 3).astype(np.int32)
        a = ((b[:, 0] << 8) | (b[:, 1] << 16) | (b[:, 2] << 24)) >> 8
    else:
        raise ValueError("unsupported sample size (8-, 16- and 24-bit PCM)")
    return np.ascontiguousarray(a.reshape(-1, w.numchannels).T)


def sample_bytes(pcm: np.ndarray, bits: int) -> bytes:
    """Wav::WriteSamples packing (file/wav.cpp:124-160): int32 [nch, n] -> interleaved little-endian sample bytes."""
    pcm = np.asarray(pcm)
    csize = (bits + 7) // 8
    if csize == 1:
        return ((pcm.T + 128) & 0xFF).astype(np.uint8).tobytes()
    if csize == 2:
        return pcm.T.astype("<i2").tobytes()
    raw = pcm.T.astype("<i4").tobytes()
    return np.frombuffer(raw, np.uint8).reshape(-1, 4)[:, :csize].tobytes()


def wav_bytes_from_pcm(pcm: np.ndarray, rate: int, bits: int = 16, extra_chunks=()) -> bytes:
    """A plain PCM WAV (tests / synthetic inputs)."""
    pcm = np.asarray(pcm)
    nch, n = pcm.shape
    data = sample_bytes(pcm, bits)
    csize = (bits + 7) // 8
    fmt = struct.pack("<HHIIHH", 1, nch, rate, rate * nch * csize, nch * csize, bits)
    body = b"WAVE" + struct.pack("<II", ID_FMT, 16) + fmt
    for cid, payload in extra_chunks:
        body += struct.pack("<II", cid, len(payload)) + payload + (b"\0" if len(payload) & 1 else b"")
    body += struct.pack("<II", ID_DATA, len(data)) + data + (b"\0" if len(data) & 1 else b"")
    return b"RIFF" + struct.pack("<I", len(body)) + body


def rebuild_wav(chunks, data: bytes) -> bytes:
    """Wav::WriteHeader + samples + pad + remaining chunks (Codec::DecodeFile, libsac.cpp:856-882)."""
    out, i = b"", 0
    while i < len(chunks):
        cid, csz, payload = chunks[i]; i += 1
        out += struct.pack("<II", cid, csz)
        if cid == ID_DATA:
            break
        out += payload
    out += data + (b"\0" if len(data) & 1 else b"")
    for cid, csz, payload in chunks[i:]:
        out += struct.pack("<II", cid, csz) + payload
    return out


def sac_header(w: WavInfo, max_framelen: int) -> bytes:
    """Sac::WriteSACHeader (file/sac.cpp:15-38)."""
    meta = pack_metadata(w.chunks)
    return (b"SAC2" + struct.pack("<HIHI", w.numchannels, w.samplerate, w.bitspersample, w.numsamples) +
            bytes([max_framelen & 0xFF, 0]) + struct.pack("<I", len(meta)) + meta)


def write_sac(path: str, w: WavInfo, max_framelen: int, records) -> int:
    """header | MD5(sample bytes) | frame records.  Returns the file size."""
    blob = sac_header(w, max_framelen) + hashlib.md5(w.data).digest() + b"".join(records)
    with open(path, "wb") as f:
        f.write(blob)
    return len(blob)


def split_records(blob: bytes, nch: int):
    """Frame records of a .sac payload (FrameCoder::ReadEncoded layout, libsac.cpp:580-600)."""
    recs, pos = [], 0
    while pos < len(blob):
        p = pos + 4 + 58 * 4
        for _ in range(nch):
            p += 18 + struct.unpack_from("<I", blob, p)[0]
        recs.append(blob[pos:p])
        pos = p
    return recs


def read_sac(path: str):
    """-> (dict header, md5 bytes, chunks, [frame records])."""
    raw = open(path, "rb").read()
    if raw[:4] != b"SAC2":
        raise ValueError("not a SAC2 file")
    nch, rate, bits, ns = struct.unpack_from("<HIHI", raw, 4)
    max_framelen = raw[16]
    msz = struct.unpack_from("<I", raw, 18)[0]
    meta = raw[22: 22 + msz]
    md5 = raw[22 + msz: 38 + msz]
    hdr = dict(numchannels=nch, samplerate=rate, bitspersample=bits, numsamples=ns, max_framelen=max_framelen, metadatasize=msz)
    return hdr, md5, unpack_metadata(meta), split_records(raw[38 + msz:], nch)


def encode_wav_files(ctx, wav_blobs, cfg, max_framelen: int = 20, adapt_block: bool = True):
    """Encode several WAV files (bytes) with one GPU context: all frames of all files form the
    batches.  Returns [(WavInfo, [frame records])]."""
    infos = [parse_wav(b) for b in wav_blobs]
    rates = {w.samplerate for w in infos}
    if len(rates) != 1 or {w.numchannels for w in infos} != {ctx.nch}:
        raise ValueError("one context encodes files of one sample rate / channel count")
    pcms = [pcm_from_wav(w) for w in infos]
    recs, _ = ctx.encode_pcm_files(pcms, rates.pop(), cfg, max_framelen=max_framelen, adapt_block=adapt_block)
    return list(zip(infos, recs))
