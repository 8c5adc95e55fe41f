"""Reader (and writer, for tests and conversions) of MindSpore ``.ckpt`` files, so that the checkpoints the reference's
CLIs load with ``ms.load_checkpoint`` (stablediffusionv2/txt2img.py:52-60, wukong-huahua/txt2img.py:60-68) drop into
``load_state_dict`` without MindSpore installed.  SURVEY.md 8(f) item 4.

Format (MindSpore ``mindspore/ccsrc/utils/checkpoint.proto``; this file restates the published schema, the reference
tree ships neither the schema nor a checkpoint -- UNPINNED against a real file):

    message Checkpoint  { repeated Value value = 1; }
    message Value       { required string tag = 1; required TensorProto tensor = 2; }
    message TensorProto { repeated int64 dims = 1; required string tensor_type = 2; required bytes tensor_content = 3; }

``ms.save_checkpoint`` writes one serialized ``Checkpoint`` per parameter -- several for a parameter larger than its slice
size, all with the same tag -- back to back; concatenated protobuf messages merge, so the file parses as ONE stream of
``value`` fields and same-tag payloads are concatenated in order.  Only the wire format is needed (varints and
length-delimited fields), so there is no protobuf dependency.
"""
import numpy as np

_DTYPES = {"Float32": np.float32, "Float16": np.float16, "Float64": np.float64, "Int8": np.int8, "Int16": np.int16,
           "Int32": np.int32, "Int64": np.int64, "UInt8": np.uint8, "UInt16": np.uint16, "UInt32": np.uint32,
           "UInt64": np.uint64, "Bool": np.bool_}
_NAMES = {np.dtype(v).name: k for k, v in _DTYPES.items()}


def _varint(buf, pos):
    shift = result = 0
    while True:
        b = buf[pos]
        pos += 1
        result |= (b & 0x7F) << shift
        if not b & 0x80:
            return result, pos
        shift += 7
        if shift > 70:
            raise ValueError("malformed varint")


def _fields(buf, pos, end):
    """Yield (field number, wire type, value) of one message; value = int for varints, memoryview for bytes."""
    while pos < end:
        key, pos = _varint(buf, pos)
        num, wt = key >> 3, key & 7
        if wt == 0:
            val, pos = _varint(buf, pos)
        elif wt == 2:
            n, pos = _varint(buf, pos)
            val = buf[pos:pos + n]
            pos += n
        elif wt == 1:
            val, pos = buf[pos:pos + 8], pos + 8
        elif wt == 5:
            val, pos = buf[pos:pos + 4], pos + 4
        else:
            raise ValueError(f"unsupported protobuf wire type {wt}")
        if pos > end:
            raise ValueError("truncated protobuf field")
        yield num, wt, val


def _signed64(v):
    return v - (1 << 64) if v >= (1 << 63) else v


def load_checkpoint(path, strip_prefix=None):
    """-> {parameter name: numpy array}.  bfloat16 payloads are widened to float32.  `strip_prefix` keeps only the names
    that start with it and removes it (e.g. "model.diffusion_model." for UNetModel.load_state_dict)."""
    with open(path, "rb") as f:
        buf = memoryview(f.read())
    chunks, meta, order = {}, {}, []
    for num, wt, val in _fields(buf, 0, len(buf)):
        if num != 1 or wt != 2:
            continue                                     # unknown top-level field: skip
        tag, dims, ttype, content = None, [], None, None
        for n2, w2, v2 in _fields(val, 0, len(val)):
            if n2 == 1 and w2 == 2:
                tag = bytes(v2).decode("utf-8")
            elif n2 == 2 and w2 == 2:
                for n3, w3, v3 in _fields(v2, 0, len(v2)):
                    if n3 == 1 and w3 == 0:
                        dims.append(_signed64(v3))
                    elif n3 == 1 and w3 == 2:            # packed repeated int64
                        p = 0
                        while p < len(v3):
                            d, p = _varint(v3, p)
                            dims.append(_signed64(d))
                    elif n3 == 2 and w3 == 2:
                        ttype = bytes(v3).decode("utf-8")
                    elif n3 == 3 and w3 == 2:
                        content = v3
        if tag is not None and strip_prefix is not None and not tag.startswith(strip_prefix):
            continue                                     # not asked for: neither decoded nor validated
        if tag is None or ttype is None or content is None:
            raise ValueError(f"{path}: checkpoint entry without tag / tensor_type / tensor_content")
        if tag not in chunks:
            chunks[tag], meta[tag] = [], (tuple(dims), ttype)
            order.append(tag)
        chunks[tag].append(content)
    out = {}
    for tag in order:
        dims, ttype = meta[tag]
        raw = b"".join(bytes(c) for c in chunks[tag])
        if ttype == "BFloat16":
            arr = (np.frombuffer(raw, dtype=np.uint16).astype(np.uint32) << 16).view(np.float32)
        elif ttype in _DTYPES:
            arr = np.frombuffer(raw, dtype=_DTYPES[ttype])
        else:
            raise ValueError(f"{path}: parameter {tag!r} has unsupported tensor_type {ttype!r}")
        n = int(np.prod(dims)) if dims else 1
        if arr.size != n:
            raise ValueError(f"{path}: parameter {tag!r}: {arr.size} elements for dims {dims}")
        name = tag if strip_prefix is None else tag[len(strip_prefix):]
        out[name] = arr.reshape(dims).copy()
    return out


def _enc_varint(v):
    v &= (1 << 64) - 1
    out = bytearray()
    while True:
        b = v & 0x7F
        v >>= 7
        out.append(b | (0x80 if v else 0))
        if not v:
            return bytes(out)


def _ld(num, payload):
    return _enc_varint((num << 3) | 2) + _enc_varint(len(payload)) + payload


def save_checkpoint(params, path, slice_bytes=512 << 20):
    """Write {name: array} in the layout ``ms.save_checkpoint`` produces (one Checkpoint message per parameter slice)."""
    with open(path, "wb") as f:
        for name, arr in params.items():
            arr = np.asarray(arr)       # (ascontiguousarray would turn a 0-d scalar into shape (1,))
            ttype = _NAMES.get(arr.dtype.name)
            if ttype is None:
                raise ValueError(f"{name}: dtype {arr.dtype} has no MindSpore tensor_type")
            raw = arr.tobytes(order="C")
            dims = b"".join(_enc_varint((1 << 3) | 0) + _enc_varint(int(d)) for d in arr.shape)
            for off in range(0, max(len(raw), 1), slice_bytes):
                tensor = dims + _ld(2, ttype.encode()) + _ld(3, raw[off:off + slice_bytes])
                value = _ld(1, name.encode("utf-8")) + _ld(2, tensor)
                f.write(_ld(1, value))


# Top-level prefixes of the reference's LatentDiffusion checkpoint (ddpm.py:75,350)
UNET_PREFIX = "model.diffusion_model."
VAE_PREFIX = "first_stage_model."
TEXT_PREFIX = "cond_stage_model."


def load_latent_diffusion(path):
    """Split one LatentDiffusion checkpoint into the three state dicts our mirrors take:
    (UNetModel.load_state_dict, AutoencoderKL.load_state_dict, FrozenCLIPEmbedder_ZH.load_state_dict)."""
    allp = load_checkpoint(path)
    pick = lambda pre: {k[len(pre):]: v for k, v in allp.items() if k.startswith(pre)}
    return pick(UNET_PREFIX), pick(VAE_PREFIX), pick(TEXT_PREFIX)
