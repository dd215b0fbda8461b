"""TFRecord files and tf.train.Example messages without TensorFlow.

The reference stores every utterance as one TFRecord file holding one serialized
tf.train.Example (nabu/processing/tfwriters/tfwriter.py:30-45); TensorFlow is not available
on the MI355X hosts, so the two public formats are implemented here from their specifications:

* TFRecord framing (tensorflow/core/lib/io/record_writer.cc): per record
  uint64 length | uint32 masked_crc32c(length) | data | uint32 masked_crc32c(data), little endian,
  mask(c) = ((c >> 15) | (c << 17)) + 0xa282ead8 (mod 2^32), CRC-32C (Castagnoli);
* Example (tensorflow/core/example/{example,feature}.proto), protobuf wire format:
  Example{1: Features}, Features{1: map<string, Feature>}, Feature{oneof 1: BytesList{1: repeated
  bytes}, 2: FloatList{1: packed float}, 3: Int64List{1: packed varint}}.
Only what the reference's writers produce and its readers parse is supported."""
import struct

# ----------------------------------------------------------------------------- CRC-32C
_TABLE = []
for _i in range(256):
    _c = _i
    for _ in range(8):
        _c = (_c >> 1) ^ 0x82F63B78 if _c & 1 else _c >> 1
    _TABLE.append(_c)


def crc32c_py(data):
    c = 0xFFFFFFFF
    for b in data:
        c = _TABLE[(c ^ b) & 0xFF] ^ (c >> 8)
    return c ^ 0xFFFFFFFF


_native = []


def crc32c(data):
    '''CRC-32C through the library's host entry point (nabu_crc32c_host, slicing-by-8, GB/s);
    the byte loop above is only the fallback of a checkout whose library has not been built —
    this is file IO, not part of the compute path'''
    if not _native:
        try:
            from nabu_amd import _hip
            _native.append(_hip.lib().nabu_crc32c_host)
        except Exception:            # library not built / not loadable
            _native.append(None)
    if _native[0] is None:
        return crc32c_py(data)
    data = bytes(data)
    return _native[0](data, len(data), 0)


def masked_crc(data):
    c = crc32c(data)
    return ((((c >> 15) | (c << 17)) & 0xFFFFFFFF) + 0xa282ead8) & 0xFFFFFFFF


# ----------------------------------------------------------------------------- framing
def write_records(path, payloads):
    with open(path, 'wb') as fid:
        for p in payloads:
            head = struct.pack('<Q', len(p))
            fid.write(head + struct.pack('<I', masked_crc(head)) + p + struct.pack('<I', masked_crc(p)))


def read_records(path, check_crc=True):
    out = []
    with open(path, 'rb') as fid:
        while True:
            head = fid.read(8)
            if not head:
                break
            if len(head) != 8:
                raise Exception('%s: truncated TFRecord header' % path)
            (n,) = struct.unpack('<Q', head)
            (hc,) = struct.unpack('<I', fid.read(4))
            data = fid.read(n)
            tail = fid.read(4)
            if len(data) != n or len(tail) != 4:
                raise Exception('%s: truncated TFRecord' % path)
            if check_crc and (hc != masked_crc(head) or struct.unpack('<I', tail)[0] != masked_crc(data)):
                raise Exception('%s: corrupted TFRecord (crc mismatch)' % path)
            out.append(data)
    return out


def peek_single_bytes_feature(path, name, head=96):
    """Byte length of the only value of BytesList feature `name` in the first record of `path`, read from
    the first `head` bytes of the file (framing + the nested length prefixes in front of the payload) —
    the sequence length of an audio-feature utterance without reading its frames.  None when the record
    does not start with that feature (several features in another order, a short file, another type)."""
    with open(path, 'rb') as fid:
        buf = fid.read(12 + head)
    if len(buf) < 16:
        return None
    try:
        pos = 12                                  # uint64 length + masked crc of the length
        for field in (1, 1, 1):                   # Example.features, Features.feature (map entry), then the key
            key, pos = _read_varint(buf, pos)
            if key != ((field << 3) | 2):
                return None
            n, pos = _read_varint(buf, pos)
        if buf[pos:pos + n] != name.encode():     # entry.key
            return None
        pos += n
        for field in (2, 1, 1):                   # entry.value = Feature, Feature.bytes_list, BytesList.value[0]
            key, pos = _read_varint(buf, pos)
            if key != ((field << 3) | 2):
                return None
            n, pos = _read_varint(buf, pos)
        return n
    except IndexError:
        return None


# ----------------------------------------------------------------------------- protobuf wire format
def _varint(n):
    n &= (1 << 64) - 1                      # int64 two's complement
    out = bytearray()
    while True:
        b = n & 0x7F
        n >>= 7
        if n:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def _read_varint(buf, pos):
    shift, val = 0, 0
    while True:
        b = buf[pos]
        pos += 1
        val |= (b & 0x7F) << shift
        if not b & 0x80:
            return val, pos
        shift += 7


def _ld(field, payload):                    # length-delimited field
    return _varint((field << 3) | 2) + _varint(len(payload)) + payload


def _fields(buf):
    """[(field number, wire type, value)] of one message"""
    pos, out = 0, []
    while pos < len(buf):
        key, pos = _read_varint(buf, pos)
        field, wt = key >> 3, key & 7
        if wt == 0:
            val, pos = _read_varint(buf, pos)
        elif wt == 2:
            n, pos = _read_varint(buf, pos)
            val = bytes(buf[pos:pos + n])
            pos += n
        elif wt == 5:
            val = bytes(buf[pos:pos + 4])
            pos += 4
        elif wt == 1:
            val = bytes(buf[pos:pos + 8])
            pos += 8
        else:
            raise Exception('unsupported protobuf wire type %d' % wt)
        out.append((field, wt, val))
    return out


def encode_example(features):
    """features: dict name -> bytes | list of bytes (BytesList), list of int (Int64List) or list of
    float (FloatList).  Map entries are written in sorted key order (what the C++ serializer of
    TF 1.8's protobuf does for deterministic output is unspecified; any order parses)."""
    entries = b''
    for key in sorted(features):
        v = features[key]
        if isinstance(v, (bytes, bytearray)):
            v = [bytes(v)]
        v = list(v)
        if v and isinstance(v[0], (bytes, bytearray)):
            feat = _ld(1, b''.join(_ld(1, bytes(x)) for x in v))
        elif v and isinstance(v[0], float):
            feat = _ld(2, _ld(1, struct.pack('<%df' % len(v), *v)))
        else:
            feat = _ld(3, _ld(1, b''.join(_varint(int(x)) for x in v)))
        entries += _ld(1, _ld(1, key.encode()) + _ld(2, feat))
    return _ld(1, entries)


def decode_example(buf):
    """-> dict name -> list of bytes | list of int | list of float"""
    out = {}
    for f, _, features in _fields(buf):
        if f != 1:
            continue
        for f2, _, entry in _fields(features):
            if f2 != 1:
                continue
            key, feat = None, b''
            for f3, _, v in _fields(entry):
                if f3 == 1:
                    key = v.decode()
                elif f3 == 2:
                    feat = v
            vals = []
            for kind, _, lst in _fields(feat):
                for f5, wt, v in _fields(lst):
                    if f5 != 1:
                        continue
                    if kind == 1:
                        vals.append(v)
                    elif kind == 2:
                        vals += list(struct.unpack('<%df' % (len(v) // 4), v)) if wt == 2 else \
                            list(struct.unpack('<f', v))
                    elif kind == 3:
                        if wt == 2:                        # packed
                            pos = 0
                            while pos < len(v):
                                x, pos = _read_varint(v, pos)
                                vals.append(x - (1 << 64) if x >> 63 else x)
                        else:
                            vals.append(v - (1 << 64) if v >> 63 else v)
            out[key] = vals
    return out
