"""Code-dataset wire format of ``viewformer-cli generate-codes`` without TensorFlow.

What the reference writes (``viewformer/data/tfrecord_dataset.py:277-331``, ``data/_common.py:241-262,383-425``,
``commands/generate_codes.py:21-75``) and its trainer reads back (``tfrecord_dataset.py:222-274``):

* ``<out>/<name>-<split>-<id:06d>-of-<n:06d>.tfrecord`` — TFRecord framing (uint64 LE length, masked CRC32C of the
  length, payload, masked CRC32C of the payload); each payload is a ``tf.train.Example`` with
  ``codes`` = Int64List (S*h*w token indices, row-major) and ``cameras`` = FloatList (S*7: xyz + quaternion);
* ``<same stem>.index`` — one ``"<offset> <record bytes>"`` line per record (``build_shard_index`` :281-297);
* ``<out>/<name>-<split>.index`` — ``"<shard id:06d> <images in sequence>"`` per sequence (``build_index`` :257-261);
* ``<out>/info.json`` — the dataset info with ``features = ['codes', 'cameras']``, ``format = 'tf'`` and
  ``token_image_size`` (``LatentCodeTransformer.update_dataset_info`` generate_codes.py:28-31).

Everything here is host-side byte work (numpy + the standard library): the protobuf wire encoding of exactly the three
message types involved is written out by hand and checked in ``tests/test_codes_dataset.py`` against the official protobuf
runtime (schema of tensorflow/core/example/{example,feature}.proto rebuilt from descriptors) and the CRC32C known answers.
The GPU only enters through the codebook model passed to :func:`generate_codes`.
"""
import json
import os
import struct
from typing import Dict, Iterable, Iterator, List, Optional, Sequence

import numpy as np

# ------------------------------------------------------------------------------------------------ CRC32C (Castagnoli)
_POLY = 0x82F63B78


def _make_tables():
    t0 = np.zeros(256, dtype=np.uint32)
    for i in range(256):
        c = i
        for _ in range(8):
            c = (c >> 1) ^ (_POLY if c & 1 else 0)
        t0[i] = c
    tables = [t0]
    for _ in range(7):                       # slicing-by-8
        prev = tables[-1]
        tables.append((prev >> 8) ^ t0[prev & 0xFF])
    return [t.tolist() for t in tables]


_T = _make_tables()


def crc32c(data: bytes, crc: int = 0) -> int:
    """CRC-32C (iSCSI polynomial, reflected), as TFRecord uses (tensorflow/core/lib/hash/crc32c.h)."""
    crc ^= 0xFFFFFFFF
    mv = memoryview(data)
    n = len(mv)
    i = 0
    t0, t1, t2, t3, t4, t5, t6, t7 = _T
    while n - i >= 8:
        lo = crc ^ (mv[i] | (mv[i + 1] << 8) | (mv[i + 2] << 16) | (mv[i + 3] << 24))
        crc = (t7[lo & 0xFF] ^ t6[(lo >> 8) & 0xFF] ^ t5[(lo >> 16) & 0xFF] ^ t4[lo >> 24]
               ^ t3[mv[i + 4]] ^ t2[mv[i + 5]] ^ t1[mv[i + 6]] ^ t0[mv[i + 7]])
        i += 8
    while i < n:
        crc = t0[(crc ^ mv[i]) & 0xFF] ^ (crc >> 8)
        i += 1
    return crc ^ 0xFFFFFFFF


def masked_crc32c(data: bytes) -> int:
    """TFRecord's masked CRC: rotate right by 15 and add a constant (tensorflow/core/lib/hash/crc32c.h: Mask)."""
    c = crc32c(data)
    return (((c >> 15) | (c << 17)) + 0xA282EAD8) & 0xFFFFFFFF


# ------------------------------------------------------------------------------------------------ protobuf wire format
def _varint(v: int) -> bytes:
    v &= (1 << 64) - 1                        # int64: negative values take the 10-byte two's complement form
    out = bytearray()
    while True:
        b = v & 0x7F
        v >>= 7
        if v:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def _read_varint(buf, pos):
    shift = result = 0
    while True:
        b = buf[pos]
        pos += 1
        result |= (b & 0x7F) << shift
        if not b & 0x80:
            return result, pos
        shift += 7
        if shift > 63:
            raise ValueError('malformed varint')


def _ld(field: int, payload: bytes) -> bytes:
    """length-delimited field"""
    return _varint((field << 3) | 2) + _varint(len(payload)) + payload


def _encode_feature(value) -> bytes:
    """tf.train.Feature: oneof { BytesList bytes_list = 1; FloatList float_list = 2; Int64List int64_list = 3 }"""
    if isinstance(value, (list, tuple)) and (len(value) == 0 or isinstance(value[0], (bytes, bytearray))):
        return _ld(1, b''.join(_ld(1, bytes(v)) for v in value))
    arr = np.asarray(value)
    if arr.dtype.kind == 'f':
        packed = np.ascontiguousarray(arr.reshape(-1), dtype='<f4').tobytes()
        return _ld(2, _ld(1, packed) if packed else b'')             # FloatList.value: packed repeated float
    if arr.dtype.kind in 'iub':
        flat = arr.reshape(-1).astype(np.int64)
        packed = b''.join(_varint(int(v)) for v in flat)
        return _ld(3, _ld(1, packed) if packed else b'')             # Int64List.value: packed repeated int64
    raise TypeError(f'unsupported feature dtype {arr.dtype}')


def encode_example(features: Dict[str, object]) -> bytes:
    """``tf.train.Example(features=Features(feature={...})).SerializeToString()`` with map keys in sorted order
    (protobuf's deterministic serialisation).  int arrays -> Int64List, float arrays -> FloatList, list of bytes -> BytesList."""
    entries = b''
    for key in sorted(features):
        entry = _ld(1, key.encode('utf-8')) + _ld(2, _encode_feature(features[key]))
        entries += _ld(1, entry)                                      # Features.feature map entry
    return _ld(1, entries)                                            # Example.features


def _fields(buf) -> Iterator:
    pos, n = 0, len(buf)
    while pos < n:
        tag, pos = _read_varint(buf, pos)
        field, wt = tag >> 3, tag & 7
        if wt == 2:
            ln, pos = _read_varint(buf, pos)
            yield field, wt, buf[pos:pos + ln]
            pos += ln
        elif wt == 0:
            v, pos = _read_varint(buf, pos)
            yield field, wt, v
        elif wt == 5:
            yield field, wt, buf[pos:pos + 4]
            pos += 4
        elif wt == 1:
            yield field, wt, buf[pos:pos + 8]
            pos += 8
        else:
            raise ValueError(f'unsupported wire type {wt}')


def _decode_feature(buf):
    for field, wt, val in _fields(buf):
        if field == 1:                                                # BytesList
            return [bytes(v) for f, _, v in _fields(val) if f == 1]
        if field == 2:                                                # FloatList: packed or one fixed32 per element
            chunks = []
            for f, w, v in _fields(val):
                if f == 1:
                    chunks.append(np.frombuffer(bytes(v), dtype='<f4'))
            return np.concatenate(chunks).astype(np.float32) if chunks else np.zeros((0,), np.float32)
        if field == 3:                                                # Int64List: packed or one varint per element
            out = []
            for f, w, v in _fields(val):
                if f != 1:
                    continue
                if w == 0:
                    out.append(v)
                else:
                    pos = 0
                    while pos < len(v):
                        x, pos = _read_varint(v, pos)
                        out.append(x)
            a = np.array(out, dtype=np.uint64).astype(np.int64) if out else np.zeros((0,), np.int64)
            return a
    return None                                                       # empty Feature


def decode_example(buf: bytes) -> Dict[str, object]:
    out = {}
    mv = memoryview(buf)
    for field, _, features in _fields(mv):
        if field != 1:
            continue
        for f, _, entry in _fields(features):
            if f != 1:
                continue
            key, value = None, None
            for ef, _, ev in _fields(entry):
                if ef == 1:
                    key = bytes(ev).decode('utf-8')
                elif ef == 2:
                    value = _decode_feature(ev)
            out[key] = value
    return out


# ------------------------------------------------------------------------------------------------ TFRecord framing
class TFRecordWriter:
    def __init__(self, path: str):
        self._f = open(path, 'wb')

    def write(self, record: bytes):
        header = struct.pack('<Q', len(record))
        self._f.write(header)
        self._f.write(struct.pack('<I', masked_crc32c(header)))
        self._f.write(record)
        self._f.write(struct.pack('<I', masked_crc32c(record)))

    def close(self):
        self._f.close()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


def read_tfrecord(path: str, check_crc: bool = True) -> Iterator[bytes]:
    with open(path, 'rb') as f:
        while True:
            header = f.read(8)
            if len(header) == 0:
                return
            if len(header) != 8:
                raise IOError(f'{path}: truncated record header')
            (length,) = struct.unpack('<Q', header)
            (hcrc,) = struct.unpack('<I', f.read(4))
            data = f.read(length)
            tail = f.read(4)
            if len(data) != length or len(tail) != 4:
                raise IOError(f'{path}: truncated record')
            if check_crc and (hcrc != masked_crc32c(header) or struct.unpack('<I', tail)[0] != masked_crc32c(data)):
                raise IOError(f'{path}: corrupted record (CRC mismatch)')
            yield data


def build_shard_index(tfrecord_file: str, index_file: str) -> None:
    """one "<offset> <bytes>" line per record (tfrecord_dataset.py:281-297)"""
    with open(tfrecord_file, 'rb') as inf, open(index_file, 'w') as out:
        while True:
            start = inf.tell()
            header = inf.read(8)
            if len(header) == 0:
                break
            (length,) = struct.unpack('<q', header)
            inf.seek(4 + length + 4, os.SEEK_CUR)
            out.write(f'{start} {inf.tell() - start}\n')


def shard_filename(path: str, split: str, shard_id: int, size: int) -> str:
    return f'{path}-{split}-{shard_id:06d}-of-{size:06d}.tfrecord'


def write_shard(stem: str, sequences: Iterable[Dict[str, object]], features: Sequence[str] = ('codes', 'cameras')) -> int:
    """``write_shard`` for code datasets (tfrecord_dataset.py:300-331): ``<stem>.tfrecord`` + ``<stem>.index``.
    Each sequence is ``dict(codes=[S,h,w] int, cameras=[S,7] float)``.  Returns the number of records."""
    n = 0
    tmp = f'{stem}.tfrecord.tmp'
    with TFRecordWriter(tmp) as w:
        for seq in sequences:
            feat = {}
            if 'cameras' in features or 'cameras-gqn' in features:
                feat['cameras'] = np.asarray(_to_numpy(seq['cameras']), dtype=np.float32).reshape(-1)
            if 'codes' in features:
                feat['codes'] = np.asarray(_to_numpy(seq['codes'])).astype(np.int64).reshape(-1)
            w.write(encode_example(feat))
            n += 1
    build_shard_index(tmp, f'{stem}.index')
    os.replace(tmp, f'{stem}.tfrecord')
    return n


def _to_numpy(x):
    return x.detach().cpu().numpy() if hasattr(x, 'detach') else np.asarray(x)


# ------------------------------------------------------------------------------------------------ dataset info / index
def write_dataset_info(path: str, dataset_info: dict, allow_incompatible_config: bool = False) -> None:
    """``write_dataset_info`` (data/_common.py:241-254): merge into an existing info.json, refuse silent config changes."""
    info = {}
    if os.path.exists(path):
        with open(path) as f:
            info = json.load(f)
    orig = dict(info)
    info.update(dataset_info)
    if not allow_incompatible_config:
        for key, val in orig.items():
            if info[key] != val and key != 'splits':
                raise RuntimeError('Cannot override dataset because dataset config is different:\n'
                                   f'{json.dumps(orig, sort_keys=True)}\n!=\n{json.dumps(info, sort_keys=True)}')
    info['splits'] = sorted(set(dataset_info['splits'] + orig.get('splits', [])))
    with open(path, 'w+') as f:
        json.dump(info, f, sort_keys=True)


def get_dataset_info(path: str) -> dict:
    with open(os.path.join(path, 'info.json')) as f:
        return json.load(f)


def _shard_map(num_images_per_sequence: Sequence[int], max_sequences_per_shard: int):
    """(sequences, images, first sequence) per shard — the max_sequences_per_shard branch of ``_get_shard_map``"""
    out, n = [], len(num_images_per_sequence)
    for off in range(0, n, max_sequences_per_shard):
        k = min(max_sequences_per_shard, n - off)
        out.append((k, int(sum(num_images_per_sequence[off:off + k])), off))
    return out


def generate_codes(sequences: Sequence[Dict[str, np.ndarray]], output_path: str, model, split: str = 'train',
                   max_sequences_per_shard: int = 1024, batch_size: int = 64, shards: Optional[List[int]] = None) -> dict:
    """``viewformer-cli generate-codes`` for in-memory sequences: encode every frame with the codebook ``model``
    (``model.encode(x)[-1]``, generate_codes.py:64-67) and write the code dataset the reference trainer consumes.

    ``sequences``: items ``dict(frames=[S,H,W,3] uint8 (or float NCHW in [-1,1]), cameras=[S,7])``.
    ``output_path``: ``<dir>/<dataset name>`` exactly as the reference's ``output`` argument.  ``shards``: 1-based shard ids
    this process writes (``--shards``, one process per GPU shards the dataset this way; shard 1's writer also writes
    info.json and the split index)."""
    import torch
    dataset_dir, name = os.path.split(output_path)
    os.makedirs(dataset_dir or '.', exist_ok=True)
    num_images = [int(len(s['frames'])) for s in sequences]
    shard_seqs = _shard_map(num_images, max_sequences_per_shard)
    n_shards = len(shard_seqs)
    first = sequences[0]['frames']
    frame_size = int(first.shape[-2])
    seq_sizes = set(num_images)
    info = {
        'name': name, 'format': 'tf', 'features': ['codes', 'cameras'], 'splits': [split],
        'frame_size': frame_size, 'num_image_channels': 3,
        'token_image_size': frame_size // int(model.config.stride),
        f'{split}_sequence_size': (num_images[0] if len(seq_sizes) == 1 else None),
        f'{split}_size': n_shards,
        f'{split}_max_images_per_shard': None, f'{split}_max_sequences_per_shard': max_sequences_per_shard,
        f'{split}_num_images': int(sum(num_images)), f'{split}_num_sequences': len(sequences),
    }
    if len({s[0] for s in shard_seqs}) == 1:
        info[f'{split}_num_sequences_per_shard'] = shard_seqs[0][0]
    if len({s[1] for s in shard_seqs}) == 1:
        info[f'{split}_num_images_per_shard'] = shard_seqs[0][1]
    todo = [i for i in (shards or range(1, n_shards + 1)) if 1 <= i <= n_shards]
    if 1 in todo:
        write_dataset_info(os.path.join(dataset_dir, 'info.json'), info, allow_incompatible_config=True)
        with open(f'{output_path}-{split}.index', 'w+') as f:
            for shard_id, (k, _, off) in enumerate(shard_seqs):
                for s in range(off, off + k):
                    f.write(f'{shard_id + 1:06d} {num_images[s]}\n')

    def encoded(seq_ids):
        # frames of consecutive sequences are batched across sequence boundaries like the reference's unbatched_/batched_ pipe
        pend_frames, pend_owner = [], []
        results = {i: [] for i in seq_ids}
        def flush():
            if not pend_frames:
                return
            x = np.concatenate(pend_frames, 0)
            codes = model.encode(torch.from_numpy(x))[-1].detach().cpu().numpy()
            pos = 0
            for i, k in pend_owner:
                results[i].append(codes[pos:pos + k])
                pos += k
            pend_frames.clear()
            pend_owner.clear()
        count = 0
        for i in seq_ids:
            fr = np.asarray(sequences[i]['frames'])
            p = 0
            while p < len(fr):
                k = min(batch_size - count, len(fr) - p)
                pend_frames.append(fr[p:p + k])
                pend_owner.append((i, k))
                count += k
                p += k
                if count == batch_size:
                    flush()
                    count = 0
        flush()
        for i in seq_ids:
            yield dict(cameras=np.asarray(sequences[i]['cameras'], dtype=np.float32), codes=np.concatenate(results[i], 0))

    for shard_id in todo:
        k, _, off = shard_seqs[shard_id - 1]
        write_shard(shard_filename(output_path, split, shard_id, n_shards)[:-len('.tfrecord')], encoded(range(off, off + k)))
    return info


def read_code_dataset(dataset_path: str, split: str, shards: Optional[List[int]] = None) -> Iterator[Dict[str, np.ndarray]]:
    """what ``read_shards`` yields for a code dataset (tfrecord_dataset.py:222-274): ``cameras`` [S,7] float32 and
    ``codes`` [S, t, t] int64 per sequence"""
    info = get_dataset_info(dataset_path)
    size = info[f'{split}_size']
    t = info['token_image_size']
    ids = [i for i in (shards or range(1, size + 1)) if 1 <= i <= size]
    for i in ids:
        path = shard_filename(os.path.join(dataset_path, info['name']), split, i, size)
        for rec in read_tfrecord(path):
            ex = decode_example(rec)
            out = {}
            if ex.get('cameras') is not None:
                out['cameras'] = ex['cameras'].reshape(-1, 7)
            if ex.get('codes') is not None:
                out['codes'] = ex['codes'].reshape(-1, t, t)
            yield out
