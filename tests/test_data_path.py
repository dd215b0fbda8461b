"""CPU: the reference's on-disk data path without TensorFlow: TFRecord framing,
tf.train.Example wire format (cross-checked with the protobuf runtime on dynamically built
descriptors of example.proto / feature.proto), the writers and readers of
nabu/processing/tf{writers,readers}, and the bucketing input pipeline."""
import configparser
import os

import numpy as np
import pytest

from nabu_amd.processing import tfrecord, input_pipeline
from nabu_amd.processing.tfreaders import tfreader_factory
from nabu_amd.processing.tfwriters import tfwriter_factory


def test_crc32c_known_answers_and_record_framing(tmp_path):
    assert tfrecord.crc32c(b'123456789') == 0xE3069283            # RFC 3720 B.4
    assert tfrecord.crc32c(bytes(32)) == 0x8A9136AA
    assert tfrecord.crc32c(bytes([0xFF] * 32)) == 0x62A8AB43
    path = str(tmp_path / 'rec')
    payloads = [b'', b'abc', os.urandom(1000)]
    tfrecord.write_records(path, payloads)
    assert tfrecord.read_records(path) == payloads
    raw = bytearray(open(path, 'rb').read())
    raw[30] ^= 1
    open(path, 'wb').write(bytes(raw))
    with pytest.raises(Exception, match='crc'):
        tfrecord.read_records(path)


def _example_classes():
    """tf.train.Example & co. from their .proto definitions, built with the protobuf runtime"""
    from google.protobuf import descriptor_pb2, descriptor_pool, message_factory
    f = descriptor_pb2.FileDescriptorProto(name='ex.proto', package='tensorflow', syntax='proto3')
    T = descriptor_pb2.FieldDescriptorProto

    def msg(name):
        m = f.message_type.add()
        m.name = name
        return m
    m = msg('BytesList'); m.field.add(name='value', number=1, type=T.TYPE_BYTES, label=T.LABEL_REPEATED)
    m = msg('FloatList'); m.field.add(name='value', number=1, type=T.TYPE_FLOAT, label=T.LABEL_REPEATED)
    m = msg('Int64List'); m.field.add(name='value', number=1, type=T.TYPE_INT64, label=T.LABEL_REPEATED)
    m = msg('Feature')
    m.oneof_decl.add(name='kind')
    for i, (n, t) in enumerate([('bytes_list', 'BytesList'), ('float_list', 'FloatList'), ('int64_list', 'Int64List')]):
        m.field.add(name=n, number=i + 1, type=T.TYPE_MESSAGE, label=T.LABEL_OPTIONAL,
                    type_name='.tensorflow.' + t, oneof_index=0)
    m = msg('Features')
    e = m.nested_type.add(name='FeatureEntry')
    e.options.map_entry = True
    e.field.add(name='key', number=1, type=T.TYPE_STRING, label=T.LABEL_OPTIONAL)
    e.field.add(name='value', number=2, type=T.TYPE_MESSAGE, label=T.LABEL_OPTIONAL, type_name='.tensorflow.Feature')
    m.field.add(name='feature', number=1, type=T.TYPE_MESSAGE, label=T.LABEL_REPEATED,
                type_name='.tensorflow.Features.FeatureEntry')
    m = msg('Example')
    m.field.add(name='features', number=1, type=T.TYPE_MESSAGE, label=T.LABEL_OPTIONAL, type_name='.tensorflow.Features')
    pool = descriptor_pool.DescriptorPool()
    pool.Add(f)
    get = getattr(message_factory, 'GetMessageClass', None)
    if get is None:
        fac = message_factory.MessageFactory(pool)
        return fac.GetPrototype(pool.FindMessageTypeByName('tensorflow.Example'))
    return get(pool.FindMessageTypeByName('tensorflow.Example'))


def test_example_wire_format_against_the_protobuf_runtime():
    Example = _example_classes()
    feats = {'data': np.arange(6, dtype=np.float32).tobytes(), 'shape': np.array([3, 2], np.int32).tobytes(),
             'length': [17], 'neg': [-3, 2 ** 40], 'fl': [1.5, -2.25]}
    mine = tfrecord.encode_example(feats)
    ex = Example()
    ex.ParseFromString(mine)                                       # my bytes parse with real protobuf
    assert ex.features.feature['data'].bytes_list.value[0] == feats['data']
    assert list(ex.features.feature['length'].int64_list.value) == [17]
    assert list(ex.features.feature['neg'].int64_list.value) == [-3, 2 ** 40]
    assert list(ex.features.feature['fl'].float_list.value) == [1.5, -2.25]
    # and bytes produced by real protobuf decode with mine (whatever field order it picks)
    ex2 = Example()
    ex2.features.feature['data'].bytes_list.value.append(b'xyz')
    ex2.features.feature['data'].bytes_list.value.append(b'')
    ex2.features.feature['length'].int64_list.value.extend([5, -1])
    ex2.features.feature['f'].float_list.value.extend([0.5])
    got = tfrecord.decode_example(ex2.SerializeToString())
    assert got == {'data': [b'xyz', b''], 'length': [5, -1], 'f': [0.5]}
    assert tfrecord.decode_example(mine)['shape'][0] == feats['shape']


def make_dataset(root, n=23, dim=5, seed=0, eos=False, min_frames=4):
    """what the reference's `run data` leaves on disk for one feature set and one text set
    (feature/string processors + ArrayWriter/StringWriter + metadata files)"""
    rng = np.random.default_rng(seed)
    alphabet = ['a', 'b', 'c', 'd']
    fdir, tdir = os.path.join(root, 'fbank'), os.path.join(root, 'text')
    fw = tfwriter_factory.factory('array')(fdir)
    tw = tfwriter_factory.factory('string')(tdir)
    feats, texts = {}, {}
    for i in range(n):
        T = int(rng.integers(min_frames, 40))
        x = rng.normal(size=(T, dim)).astype(np.float32)
        L = int(rng.integers(1, 6))
        txt = ' '.join(alphabet[j] for j in rng.integers(0, 4, L))
        name = 'utt%02d' % i
        fw.write(x, name)
        tw.write(txt, name)
        feats[name], texts[name] = x, txt
    for d, lens in ((fdir, [v.shape[0] for v in feats.values()]), (tdir, [len(t.split()) for t in texts.values()])):
        open(os.path.join(d, 'max_length'), 'w').write(str(max(lens)))
        np.save(os.path.join(d, 'sequence_length_histogram.npy'), np.bincount(lens, minlength=max(lens) + 1))
    open(os.path.join(fdir, 'dim'), 'w').write(str(dim))
    open(os.path.join(tdir, 'alphabet'), 'w').write(' '.join(alphabet))
    open(os.path.join(tdir, 'nonesymbol'), 'w').write('<none>')
    conf = configparser.ConfigParser()
    conf.read_dict({'trainfbank': {'type': 'audio_feature', 'dir': fdir},
                    'traintext': {'type': 'string_eos' if eos else 'string', 'dir': tdir}})
    return conf, feats, texts, alphabet


def test_writers_and_readers_round_trip(tmp_path):
    conf, feats, texts, alphabet = make_dataset(str(tmp_path))
    fr = tfreader_factory.factory('audio_feature')([conf.get('trainfbank', 'dir')])
    sr = tfreader_factory.factory('string')([conf.get('traintext', 'dir')])
    er = tfreader_factory.factory('string_eos')([conf.get('traintext', 'dir')])
    assert fr.metadata['dim'] == 5 and fr.metadata['sequence_length_histogram'].sum() == 23
    assert er.metadata['eos_label'] == 4 and er.metadata['max_length'] == sr.metadata['max_length'] + 1
    elements, names = input_pipeline.get_filenames([[dict(conf.items('trainfbank'))], [dict(conf.items('traintext'))]])
    assert len(elements) == 23
    for (ff, tf_), name in zip(elements, names):
        key = name.rsplit('-', 1)[0]
        x, n = fr(ff)
        assert n == feats[key].shape[0] and np.array_equal(x, feats[key])
        y, ln = sr(tf_)
        assert list(y) == [alphabet.index(c) for c in texts[key].split()] and ln == len(y)
        ye, le = er(tf_)
        assert list(ye) == list(y) + [4] and le == ln + 1
    with pytest.raises(Exception, match='unknown data type'):
        tfreader_factory.factory('nope')


def test_bucket_boundaries_is_the_reference_greedy():
    hist = np.zeros(21)
    hist[[3, 5, 8, 13, 20]] = [10, 10, 10, 10, 10]
    assert input_pipeline.bucket_boundaries(hist, 1) == []
    b = input_pipeline.bucket_boundaries(hist, 5)
    counts = [hist[lo:hi].sum() for lo, hi in zip([0] + b, b + [21])]
    assert counts == [10, 10, 10, 10, 10]                       # one length class per bucket
    assert input_pipeline.bucket_boundaries(hist, 2) in ([9], [8 + 1], [13])   # half/half up to the greedy tie rule


def test_record_data_batches(tmp_path):
    conf, feats, texts, alphabet = make_dataset(str(tmp_path), n=40, seed=3)
    def src(**kw):
        return input_pipeline.from_sections(conf, ['features'], [['trainfbank']], ['text'], [['traintext']], **kw)
    plain = src(batch_size=4, numbuckets=1, seed=1)
    assert plain.num_batches() == 10
    b = plain.batch(0)
    assert b['inputs']['features'].shape[0] == 4 and b['inputs']['features'].dtype == np.float32
    assert b['targets']['text'].dtype == np.int32 and b['input_seq_length']['features'].dtype == np.int32
    T = b['input_seq_length']['features']
    assert b['inputs']['features'].shape[1] == T.max()
    for j in range(4):
        assert np.all(b['inputs']['features'][j, T[j]:] == 0)
    # an epoch visits every utterance exactly once, epochs are shuffled differently, access is random
    seen = np.concatenate([plain._indices(s) for s in range(10)])
    assert sorted(seen) == list(range(40))
    assert list(np.concatenate([plain._indices(s) for s in range(10, 20)])) != list(seen)
    assert np.array_equal(src(batch_size=4, numbuckets=1, seed=1).batch(7)['inputs']['features'],
                          plain.batch(7)['inputs']['features'])
    # bucketing: members of a batch share a bucket; longer buckets get smaller batches
    bk = src(batch_size=6, numbuckets=3, variable_batch_size=True, seed=2)
    lens = bk._first_lengths()
    assert bk.batch_sizes[0] == 6 and bk.batch_sizes[-1] < 6 and len(bk.batch_sizes) == 3
    for s in range(12):
        idx = bk._indices(s)
        buckets = {int(np.searchsorted(bk.boundaries, lens[u], side='right')) for u in idx}
        assert len(buckets) == 1 and len(idx) == bk.batch_sizes[buckets.pop()]


def test_trainer_and_evaluator_read_the_database_conf(tmp_path):
    from nabu_amd import recipes
    from nabu_amd.neuralnetworks.trainers import trainer_factory
    conf, feats, texts, alphabet = make_dataset(str(tmp_path / 'train'), n=24, dim=40)
    dev, _, _, _ = make_dataset(str(tmp_path / 'dev'), n=9, dim=40, seed=5)
    conf.read_dict({'devfbank': dict(dev.items('trainfbank')), 'devtext': dict(dev.items('traintext'))})
    mc, tc, ec = recipes.load_recipe('cfg2_listener_ctc', **{'trainer.batch_size': 4, 'trainer.numbuckets': 2,
                                                             'evaluator.batch_size': 2})
    tc.set('trainer', 'features', 'trainfbank')
    tc.set('trainer', 'targets', 'text')
    tc.set('trainer', 'text', 'traintext')
    tr = trainer_factory.factory('standard')(conf=tc, dataconf=conf, modelconf=mc, evaluatorconf=ec,
                                             expdir=None, server=None, task_index=0)
    out = tr._create_graph()
    assert isinstance(tr.data, input_pipeline.RecordData) and len(tr.data.batch_sizes) == 2
    assert out['num_steps'] == tr.data.num_batches() * int(tc.get('trainer', 'num_epochs'))
    b = tr.data.batch(0)
    assert set(b['inputs']) == {'features'} and set(b['targets']) == {'text'}
    assert tr.evaluator is not None and tr.evaluator.data.num_batches() == 4          # 9 // 2
    assert tr.evaluator.data.batch(0)['inputs']['features'].shape[0] == 2


def test_lengths_without_decoding_and_background_prefetch(tmp_path):
    """bucketing takes every utterance's length from the record's length prefixes (no frame is read before
    the first step), and the prefetching batch source hands out exactly the batches of a source that reads
    on demand — in order, with a stride (data-parallel ranks), and after a jump"""
    conf, feats, texts, alphabet = make_dataset(str(tmp_path), n=40, dim=7)
    fr = tfreader_factory.factory('audio_feature')([conf.get('trainfbank', 'dir')])
    elements, names = input_pipeline.get_filenames([[dict(conf.items('trainfbank'))], [dict(conf.items('traintext'))]])
    calls = {'n': 0}
    orig = type(fr).__call__

    def counting(self, filename):
        calls['n'] += 1
        return orig(self, filename)
    type(fr).__call__ = counting
    try:
        for (ff, _), name in zip(elements, names):
            assert fr.sequence_length(ff) == feats[name.rsplit('-', 1)[0]].shape[0]
        assert calls['n'] == 0                       # nothing was decoded
    finally:
        type(fr).__call__ = orig
    # a record that does not start with the 'data' feature falls back to decoding
    from nabu_amd.processing import tfrecord
    odd = str(tmp_path / 'odd.tfrecord')
    tfrecord.write_records(odd, [tfrecord.encode_example({'aaa': [1, 2], 'data': np.zeros((3, 7), np.float32).tobytes()})])
    assert tfrecord.peek_single_bytes_feature(odd, 'data') is None and fr.sequence_length(odd) == 3

    def source():
        return input_pipeline.from_sections(conf, ['features'], [['trainfbank']], ['text'], [['traintext']],
                                            batch_size=4, numbuckets=3, variable_batch_size=True, shuffle=True, seed=3)
    a, b = source(), source()
    b._pool = None
    b.batch = b._assemble                            # reads on demand, no prefetch
    for step in [0, 1, 2, 3, 5, 7, 9, 30, 31, 2]:
        x, y = a.batch(step), b.batch(step)
        for key in ('inputs', 'input_seq_length', 'targets', 'target_seq_length'):
            for n in x[key]:
                np.testing.assert_array_equal(x[key][n], y[key][n])
    assert len(a._ahead) == 1
