"""CPU: checkpoint ingestion (viewformer_amd/checkpoint.py).  The torch half is pinned by a model directory
written by the reference's own classes (tests/golden/vqgan_tiny_model, make_ckpt_golden.py); the TensorBundle half is
checked against its own writer + the published format constants (parity unpinned: no TensorFlow in the image)."""
import json
import os
import struct

import numpy as np
import pytest

from viewformer_amd import checkpoint as ck
from viewformer_amd.config import MIGTConfig, VQGANConfig, load_config

HERE = os.path.dirname(os.path.abspath(__file__))
TINY_DIR = os.path.join(HERE, 'golden', 'vqgan_tiny_model')


def test_reference_written_model_directory_loads():
    cfg = load_config(json.load(open(os.path.join(TINY_DIR, 'config.json'))))
    assert isinstance(cfg, VQGANConfig) and cfg.ch == 32 and cfg.n_embed == 64 and cfg.image_size == 32
    sd = ck.read_torch_checkpoint(os.path.join(TINY_DIR, 'model.ckpt'))
    from viewformer_amd.vqgan import VQGAN
    from viewformer_amd.weights import make_vqgan_weights
    m = VQGAN(cfg)
    m.load_state_dict(sd)                                    # strict: loss sub-module keys are ignored, nothing missing
    want = make_vqgan_weights(cfg, seed=3, codebook_scale=0.05)
    got = m.state_dict()
    assert set(got) == set(want)
    for k in want:
        assert np.array_equal(np.asarray(got[k]).reshape(-1), np.asarray(want[k]).reshape(-1)), k   # (the reference's counter is [1])
    assert sd['encoder.conv_in.weight'].shape == (32, 3, 3, 3)               # OIHW, as the reference ships them


def test_tensor_bundle_roundtrip_and_format_constants(tmp_path):
    rng = np.random.default_rng(0)
    tensors = {f'h/{i}/attn/c_attn/weight/.ATTRIBUTES/VARIABLE_VALUE': rng.normal(size=(4, 6)).astype(np.float32) for i in range(700)}
    tensors['save_counter/.ATTRIBUTES/VARIABLE_VALUE'] = np.array(3, dtype=np.int64)
    tensors['wpe/.ATTRIBUTES/VARIABLE_VALUE'] = rng.normal(size=(256, 16)).astype(np.float32)
    prefix = str(tmp_path / 'model')
    ck.write_tensor_bundle(prefix, tensors)
    raw = open(prefix + '.index', 'rb').read()
    assert struct.unpack('<Q', raw[-8:])[0] == 0xdb4775248b80fb57           # leveldb table magic
    assert os.path.getsize(prefix + '.data-00000-of-00001') == sum(t.nbytes for t in tensors.values())
    table = ck._read_table(prefix + '.index')
    assert list(table)[0] == b'' and list(table)[1:] == sorted(k.encode() for k in tensors)   # header first, keys sorted
    assert len(raw) > 2 * 4096                                               # several data blocks -> the index block is exercised
    back = ck.read_tensor_bundle(prefix)
    assert set(back) == set(tensors)
    for k, v in tensors.items():
        assert back[k].dtype == v.dtype and back[k].shape == v.shape and np.array_equal(back[k], v)
    # a flipped data byte is caught by the per-tensor CRC
    with open(prefix + '.data-00000-of-00001', 'r+b') as f:
        f.seek(100)
        b = f.read(1)
        f.seek(100)
        f.write(bytes([b[0] ^ 1]))
    with pytest.raises(IOError):
        ck.read_tensor_bundle(prefix)
    # and a flipped index byte by the block CRC
    ck.write_tensor_bundle(prefix, tensors)
    raw = bytearray(open(prefix + '.index', 'rb').read())
    raw[50] ^= 1
    open(prefix + '.index', 'wb').write(bytes(raw))
    with pytest.raises(IOError):
        ck.read_tensor_bundle(prefix)


def test_keras_key_mapping_both_conventions():
    from viewformer_amd.weights import make_migt_weights
    cfg = MIGTConfig(n_layer=2, d_model=64, n_head=1, localization_weight='1')
    sd = make_migt_weights(cfg, seed=0)
    obj = ck.state_dict_to_keras(sd)                                          # object-graph keys (model.save_weights)
    assert 'h/1/attn/c_attn/weight/.ATTRIBUTES/VARIABLE_VALUE' in obj and 'wpe/.ATTRIBUTES/VARIABLE_VALUE' in obj
    assert obj['h/0/mlp/c_fc/bias/.ATTRIBUTES/VARIABLE_VALUE'].shape == (1, 4 * 64)          # Conv1D bias is [1, nf] (migt.py:87)
    obj['optimizer/iter/.ATTRIBUTES/VARIABLE_VALUE'] = np.array(7, np.int64)
    obj['h/0/attn/c_attn/weight/.OPTIMIZER_SLOT/optimizer/m/.ATTRIBUTES/VARIABLE_VALUE'] = np.zeros((64, 192), np.float32)
    back = ck.keras_to_state_dict(obj)
    assert set(back) == set(sd)
    for k in sd:
        assert np.array_equal(back[k], np.asarray(sd[k])), k
    # name-based keys: 'migt/h.0/attn/c_attn/weight', 'migt/wte/weight', 'migt/wpe/embeddings'
    named = {}
    for k, v in sd.items():
        parts = k.split('.')
        key = ('h.' + parts[1] + '/' + '/'.join(parts[2:])) if parts[0] == 'h' else '/'.join(parts)
        named['migt/' + key] = np.asarray(v)
    back2 = ck.keras_to_state_dict(named)
    assert set(back2) == set(sd)


def test_load_model_reads_config_overrides_and_refuses_missing_dirs(tmp_path):
    from viewformer_amd.weights import make_migt_weights
    cfg = MIGTConfig(n_layer=1, d_model=64, n_head=1, localization_weight='1', pose_multiplier=1.0)
    d = tmp_path / 'tr'
    d.mkdir()
    from dataclasses import asdict
    cj = asdict(cfg)
    cj['model'] = 'migt'
    json.dump(cj, open(d / 'config.json', 'w'))
    ck.write_tensor_bundle(str(d / 'model'), ck.state_dict_to_keras(make_migt_weights(cfg, seed=1)))
    m = ck.load_model(str(d / 'model'), pose_multiplier=0.2)                 # evaluate_transformer.py:205-208
    assert m.config.pose_multiplier == 0.2 and m.config.n_layer == 1
    assert np.array_equal(np.asarray(m.state_dict()['wte.weight']), np.asarray(make_migt_weights(cfg, seed=1)['wte.weight']))
    with pytest.raises(TypeError, match='pose_multipler'):                   # a mistyped override must not be dropped silently
        ck.load_model(str(d / 'model'), pose_multipler=0.2)
    cj2 = dict(cj, some_future_field=3)                                       # ... while unknown keys INSIDE config.json are ignored
    json.dump(cj2, open(d / 'config.json', 'w'))                              # (models/__init__.py:63-74 walks the dataclass fields only)
    assert ck.load_model(str(d / 'model')).config.n_layer == 1
    with pytest.raises(FileNotFoundError):
        ck.load_model('interiornet-transformer-tf')                          # named checkpoints need the network
    # a checkpoint with a missing tensor is refused by load_state_dict
    sd = make_migt_weights(cfg, seed=1)
    sd.pop('ln_f.gamma')
    ck.write_tensor_bundle(str(d / 'broken'), ck.state_dict_to_keras(sd))
    with pytest.raises(RuntimeError):
        ck.load_model(str(d / 'broken'))


def test_optimizer_slots_key_mapping_roundtrip():
    """the optimizer half of a compiled model's TF checkpoint: object-graph slot keys <-> the trainer's optimizer state dict"""
    from viewformer_amd import checkpoint as ck
    osd = {'iterations': 1234, 'lr_offset': 7, 'm/h.0.attn.c_attn.weight': np.ones((4, 12), np.float32), 'v/h.0.attn.c_attn.bias': np.arange(12, dtype=np.float32),
           'm/wpe.embeddings': np.zeros((3, 4), np.float32), 'v/ln_f.gamma': np.full((4,), 2.0, np.float32)}
    k = ck.optimizer_state_to_keras(osd)
    assert 'h/0/attn/c_attn/weight/.OPTIMIZER_SLOT/optimizer/m/.ATTRIBUTES/VARIABLE_VALUE' in k
    assert k['h/0/attn/c_attn/bias/.OPTIMIZER_SLOT/optimizer/v/.ATTRIBUTES/VARIABLE_VALUE'].shape == (1, 12)       # Conv1D bias [1, nf] (migt.py:87)
    assert 'wpe/.OPTIMIZER_SLOT/optimizer/m/.ATTRIBUTES/VARIABLE_VALUE' in k and int(k['optimizer/iter/.ATTRIBUTES/VARIABLE_VALUE']) == 1234
    back = ck.keras_optimizer_state(k)
    assert back['iterations'] == 1234 and back['lr_offset'] == 7
    for key in osd:
        if key.startswith(('m/', 'v/')):
            assert np.array_equal(back[key], osd[key]) and back[key].shape == osd[key].shape, key
    # the weights half ignores the slots, the optimizer half ignores the weights
    both = dict(ck.state_dict_to_keras({'ln_f.gamma': np.ones(4, np.float32)}), **k)
    assert list(ck.keras_to_state_dict(both)) == ['ln_f.gamma']
    assert 'm/ln_f.gamma' not in ck.keras_optimizer_state(both) and 'v/ln_f.gamma' in ck.keras_optimizer_state(both)


def _bundle_classes():
    """BundleHeaderProto / BundleEntryProto / TensorShapeProto / VersionDef rebuilt from descriptors (tensorflow/core/protobuf/tensor_bundle.proto,
    framework/tensor_shape.proto, framework/versions.proto — message and field numbers as published) for the OFFICIAL protobuf runtime"""
    from google.protobuf import descriptor_pb2, descriptor_pool, message_factory
    fd = descriptor_pb2.FileDescriptorProto(name='vf_bundle.proto', package='tensorflow', syntax='proto3')
    F = descriptor_pb2.FieldDescriptorProto

    def msg(name, parent=None):
        m = (parent.nested_type if parent is not None else fd.message_type).add()
        m.name = name
        return m

    def field(m, name, number, ftype, label=F.LABEL_OPTIONAL, type_name=None):
        f = m.field.add(name=name, number=number, type=ftype, label=label)
        if type_name:
            f.type_name = type_name
    shape = msg('TensorShapeProto')
    dim = msg('Dim', shape)
    field(dim, 'size', 1, F.TYPE_INT64)
    field(dim, 'name', 2, F.TYPE_STRING)
    field(shape, 'dim', 2, F.TYPE_MESSAGE, F.LABEL_REPEATED, '.tensorflow.TensorShapeProto.Dim')
    field(shape, 'unknown_rank', 3, F.TYPE_BOOL)
    ver = msg('VersionDef')
    field(ver, 'producer', 1, F.TYPE_INT32)
    field(ver, 'min_consumer', 2, F.TYPE_INT32)
    hdr = msg('BundleHeaderProto')
    field(hdr, 'num_shards', 1, F.TYPE_INT32)
    field(hdr, 'endianness', 2, F.TYPE_INT32)            # (an enum on the wire is a varint)
    field(hdr, 'version', 3, F.TYPE_MESSAGE, type_name='.tensorflow.VersionDef')
    ent = msg('BundleEntryProto')
    field(ent, 'dtype', 1, F.TYPE_INT32)                 # DataType enum
    field(ent, 'shape', 2, F.TYPE_MESSAGE, type_name='.tensorflow.TensorShapeProto')
    field(ent, 'shard_id', 3, F.TYPE_INT32)
    field(ent, 'offset', 4, F.TYPE_INT64)
    field(ent, 'size', 5, F.TYPE_INT64)
    field(ent, 'crc32c', 6, F.TYPE_FIXED32)
    pool = descriptor_pool.DescriptorPool()
    pool.Add(fd)
    get = getattr(message_factory, 'GetMessageClass', None) or (lambda d: message_factory.MessageFactory(pool).GetPrototype(d))
    return get(pool.FindMessageTypeByName('tensorflow.BundleHeaderProto')), get(pool.FindMessageTypeByName('tensorflow.BundleEntryProto'))


def test_bundle_protos_match_the_protobuf_runtime(tmp_path):
    """the proto layer of the TensorBundle reader / writer against the official protobuf runtime (the SSTable layer under it and TensorFlow's own
    files remain unpinned: f2 stays partial): entries and the header as this module writes them are byte-identical to the runtime's serialisation
    of the published messages, the reader parses what the runtime writes, and the entries of a bundle written here parse with the runtime."""
    from viewformer_amd import checkpoint as ck
    Header, Entry = _bundle_classes()
    cases = [(1, (768, 2304), 0, 0, 768 * 2304 * 4, 0xdeadbeef), (9, (), 0, 4096, 8, 1), (1, (1, 3072), 2, 123456789012, 12288, 0xffffffff),
             (3, (0,), 0, 7, 0, 0)]
    for dtype, shape, shard, off, size, crc in cases:
        e = Entry(dtype=dtype, shard_id=shard, offset=off, size=size, crc32c=crc)
        for d in shape:
            e.shape.dim.add(size=d)
        if not shape:
            e.shape.SetInParent()
        official = e.SerializeToString(deterministic=True)
        mine = ck._entry_proto(dtype, shape, shard, off, size, crc)
        back = Entry.FromString(mine)
        assert (back.dtype, [d.size for d in back.shape.dim], back.shard_id, back.offset, back.size, back.crc32c) == (dtype, list(shape), shard, off, size, crc)
        p = ck._parse_entry(official)
        assert (p['dtype'], p['shape'], p['shard'], p['offset'], p['size']) == (dtype, list(shape), shard, off, size)
        assert p['crc'] == (crc if crc else None)             # (proto3 omits a zero crc: the reader then has nothing to check)
        if crc:                                            # (proto3 omits zero scalars; this writer always emits size and crc: equal bytes when non-zero)
            assert mine == official or size == 0, (mine.hex(), official.hex())
    h = Header(num_shards=1)
    h.version.producer = 1
    prefix = str(tmp_path / 'm')
    ck.write_tensor_bundle(prefix, {'a/b': np.arange(6, dtype=np.float32).reshape(2, 3), 'c': np.array(7, dtype=np.int64)})
    table = ck._read_table(prefix + '.index')
    assert table[b''] == h.SerializeToString(deterministic=True)              # the header entry, byte for byte
    for key, raw in table.items():
        if key:
            ent = Entry.FromString(raw)
            assert ent.size == {b'a/b': 24, b'c': 8}[key] and ent.dtype in (1, 9) and ent.crc32c != 0
