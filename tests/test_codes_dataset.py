"""CPU: the code-dataset wire format (viewformer_amd/codes_dataset.py) without TensorFlow.
Pins: CRC32C known answers (RFC 3720 B.4) + TFRecord's mask; the hand-written protobuf encoder/decoder against the
OFFICIAL protobuf runtime with the tf.train.Example schema rebuilt from descriptors (byte-identical serialisation and
cross-parsing); framing/index/info.json against the reference's own reader loop (tfrecord_dataset.py:281-297 restated)."""
import json
import os
import struct

import numpy as np
import pytest

from viewformer_amd import codes_dataset as cd


def test_crc32c_known_answers():
    assert cd.crc32c(b'123456789') == 0xE3069283
    assert cd.crc32c(bytes(32)) == 0x8A9136AA                       # RFC 3720 B.4
    assert cd.crc32c(b'\xff' * 32) == 0x62A8AB43
    assert cd.crc32c(bytes(range(32))) == 0x46DD794E
    assert cd.crc32c(bytes(range(31, -1, -1))) == 0x113FDB5C
    data = np.random.default_rng(0).integers(0, 256, 1000, dtype=np.uint8).tobytes()
    assert cd.crc32c(data[400:], cd.crc32c(data[:400])) == cd.crc32c(data)     # streaming form
    c = cd.crc32c(b'123456789')
    assert cd.masked_crc32c(b'123456789') == (((c >> 15) | (c << 17)) + 0xA282EAD8) & 0xFFFFFFFF


def _example_classes():
    """tf.train.{BytesList,FloatList,Int64List,Feature,Features,Example} rebuilt from descriptors
    (tensorflow/core/example/feature.proto, example.proto — field numbers and packing as published)"""
    from google.protobuf import descriptor_pb2, descriptor_pool, message_factory
    fd = descriptor_pb2.FileDescriptorProto(name='vf_example.proto', package='tensorflow', syntax='proto3')
    F = descriptor_pb2.FieldDescriptorProto

    def msg(name):
        m = fd.message_type.add()
        m.name = name
        return m

    def field(m, name, number, ftype, label=F.LABEL_OPTIONAL, type_name=None, packed=None, oneof=None):
        f = m.field.add(name=name, number=number, type=ftype, label=label)
        if type_name:
            f.type_name = type_name
        if packed is not None:
            f.options.packed = packed
        if oneof is not None:
            f.oneof_index = oneof
        return f
    field(msg('BytesList'), 'value', 1, F.TYPE_BYTES, F.LABEL_REPEATED)
    field(msg('FloatList'), 'value', 1, F.TYPE_FLOAT, F.LABEL_REPEATED, packed=True)
    field(msg('Int64List'), 'value', 1, F.TYPE_INT64, F.LABEL_REPEATED, packed=True)
    feat = msg('Feature')
    feat.oneof_decl.add(name='kind')
    field(feat, 'bytes_list', 1, F.TYPE_MESSAGE, type_name='.tensorflow.BytesList', oneof=0)
    field(feat, 'float_list', 2, F.TYPE_MESSAGE, type_name='.tensorflow.FloatList', oneof=0)
    field(feat, 'int64_list', 3, F.TYPE_MESSAGE, type_name='.tensorflow.Int64List', oneof=0)
    feats = msg('Features')
    entry = feats.nested_type.add(name='FeatureEntry')
    entry.options.map_entry = True
    field(entry, 'key', 1, F.TYPE_STRING)
    field(entry, 'value', 2, F.TYPE_MESSAGE, type_name='.tensorflow.Feature')
    field(feats, 'feature', 1, F.TYPE_MESSAGE, F.LABEL_REPEATED, type_name='.tensorflow.Features.FeatureEntry')
    field(msg('Example'), 'features', 1, F.TYPE_MESSAGE, type_name='.tensorflow.Features')
    pool = descriptor_pool.DescriptorPool()
    pool.Add(fd)
    get = getattr(message_factory, 'GetMessageClass', None)
    if get is None:
        factory = message_factory.MessageFactory(pool)
        return factory.GetPrototype(pool.FindMessageTypeByName('tensorflow.Example'))
    return get(pool.FindMessageTypeByName('tensorflow.Example'))


def test_example_encoding_matches_the_protobuf_runtime():
    Example = _example_classes()
    rng = np.random.default_rng(1)
    codes = rng.integers(0, 1024, (7, 8, 8)).astype(np.int64)
    codes[0, 0, 0], codes[0, 0, 1] = -1, 2 ** 40                     # negative (10-byte varint) and wide values survive too
    cameras = rng.normal(size=(7, 7)).astype(np.float32)
    mine = cd.encode_example({'codes': codes, 'cameras': cameras})
    ex = Example()
    ex.features.feature['codes'].int64_list.value.extend(int(v) for v in codes.reshape(-1))
    ex.features.feature['cameras'].float_list.value.extend(float(v) for v in cameras.reshape(-1))
    official = ex.SerializeToString(deterministic=True)
    assert mine == official
    # cross-parsing both ways
    back = Example.FromString(mine)
    assert list(back.features.feature['codes'].int64_list.value) == codes.reshape(-1).tolist()
    assert np.array_equal(np.array(back.features.feature['cameras'].float_list.value, dtype=np.float32), cameras.reshape(-1))
    dec = cd.decode_example(official)
    assert np.array_equal(dec['codes'], codes.reshape(-1)) and dec['codes'].dtype == np.int64
    assert np.array_equal(dec['cameras'], cameras.reshape(-1)) and dec['cameras'].dtype == np.float32
    # bytes features (frames) and empty lists
    ex2 = Example()
    ex2.features.feature['frames'].bytes_list.value.extend([b'\xff\xd8jpeg0', b'', b'x' * 300])
    ex2.features.feature['codes'].int64_list.SetInParent()
    mine2 = cd.encode_example({'frames': [b'\xff\xd8jpeg0', b'', b'x' * 300], 'codes': np.zeros((0,), np.int64)})
    assert mine2 == ex2.SerializeToString(deterministic=True)
    d2 = cd.decode_example(mine2)
    assert d2['frames'] == [b'\xff\xd8jpeg0', b'', b'x' * 300] and d2['codes'].size == 0


def test_unpacked_repeated_fields_are_accepted():
    """old writers emit one element per tag instead of the packed form; parsers must take both"""
    body = b''.join(b'\x08' + cd._varint(v) for v in (5, 300, 7))          # Int64List.value, wire type 0, one per element
    feature = cd._ld(3, body)
    entry = cd._ld(1, b'codes') + cd._ld(2, feature)
    buf = cd._ld(1, cd._ld(1, entry))
    assert cd.decode_example(buf)['codes'].tolist() == [5, 300, 7]


def _reference_index_loop(tfrecord_file):
    """restatement of build_shard_index (tfrecord_dataset.py:281-297) returning its lines"""
    lines = []
    with open(tfrecord_file, 'rb') as infile:
        while True:
            current = infile.tell()
            byte_len = infile.read(8)
            if len(byte_len) == 0:
                break
            infile.read(4)
            proto_len = struct.unpack('q', byte_len)[0]
            infile.read(proto_len)
            infile.read(4)
            lines.append(f'{current} {infile.tell() - current}')
    return lines


class _FakeCodebook:
    """stands in for the VQGAN on CPU: deterministic 'codes' from the frames (the GPU model is exercised in test_hip_models)"""
    class config:
        stride = 16
        image_size = 32

    def encode(self, x):
        import torch
        assert x.dtype == torch.uint8 and x.shape[1:] == (32, 32, 3)
        pooled = x.reshape(-1, 2, 16, 2, 16, 3).to(torch.int64).sum((2, 4, 5)) % 1024
        return None, None, pooled


def test_generate_codes_roundtrip_and_layout(tmp_path):
    rng = np.random.default_rng(3)
    lengths = [5, 5, 5, 5, 5, 5, 5]
    seqs = [dict(frames=rng.integers(0, 256, (n, 32, 32, 3), dtype=np.uint8), cameras=rng.normal(size=(n, 7)).astype(np.float32))
            for n in lengths]
    out = str(tmp_path / 'ds' / 'toy')
    model = _FakeCodebook()
    # two "processes", like --shards 1 / --shards 2,3
    cd.generate_codes(seqs, out, model, split='train', max_sequences_per_shard=3, batch_size=4, shards=[2, 3])
    info = cd.generate_codes(seqs, out, model, split='train', max_sequences_per_shard=3, batch_size=4, shards=[1])
    d = str(tmp_path / 'ds')
    files = sorted(os.listdir(d))
    assert files == ['info.json', 'toy-train-000001-of-000003.index', 'toy-train-000001-of-000003.tfrecord',
                     'toy-train-000002-of-000003.index', 'toy-train-000002-of-000003.tfrecord',
                     'toy-train-000003-of-000003.index', 'toy-train-000003-of-000003.tfrecord', 'toy-train.index']
    disk = json.load(open(os.path.join(d, 'info.json')))
    assert disk == json.loads(json.dumps(info)) | {'splits': ['train']}
    assert disk['features'] == ['codes', 'cameras'] and disk['format'] == 'tf' and disk['token_image_size'] == 2
    assert disk['train_sequence_size'] == 5 and disk['train_size'] == 3 and disk['train_num_sequences'] == 7
    assert open(os.path.join(d, 'toy-train.index')).read().split('\n')[:4] == ['000001 5', '000001 5', '000001 5', '000002 5']
    for i in (1, 2, 3):
        stem = os.path.join(d, f'toy-train-{i:06d}-of-000003')
        assert open(stem + '.index').read().strip().split('\n') == _reference_index_loop(stem + '.tfrecord')
    back = list(cd.read_code_dataset(d, 'train'))
    assert len(back) == 7
    for s, b in zip(seqs, back):
        import torch
        want = model.encode(torch.from_numpy(s['frames']))[-1].numpy()
        assert b['codes'].shape == (5, 2, 2) and np.array_equal(b['codes'], want)       # batching across sequences kept order
        assert np.array_equal(b['cameras'], s['cameras'])
    assert len(list(cd.read_code_dataset(d, 'train', shards=[3]))) == 1
    # a different config may not silently replace the dataset's
    with pytest.raises(RuntimeError):
        cd.write_dataset_info(os.path.join(d, 'info.json'), dict(disk, token_image_size=4))


def test_corruption_is_detected(tmp_path):
    p = str(tmp_path / 'x.tfrecord')
    with cd.TFRecordWriter(p) as w:
        w.write(cd.encode_example({'codes': np.arange(10)}))
    raw = bytearray(open(p, 'rb').read())
    raw[20] ^= 1
    open(p, 'wb').write(bytes(raw))
    with pytest.raises(IOError):
        list(cd.read_tfrecord(p))
    assert len(list(cd.read_tfrecord(p, check_crc=False))) == 1
