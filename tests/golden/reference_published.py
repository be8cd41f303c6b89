"""Numbers the reference tree itself publishes for this path, and the inputs they were produced on.

These are the only OUTPUTS of the reference's libFLAC path that exist under /root/reference: the executed notebooks
of its documentation print compressed sizes and restored values for arrays made by `demo.create_fake_data` with its
default seed.  They are data (three integers, six floats, the cell numbers they come from); the generator below is a
restatement of the float branch of `create_fake_data` (src/flacarray/demo.py:70-97; seed 123456789, demo.py:12),
written against its sequence of random draws -- one `random()` block for the DC levels, one for the scales, then ONE
stream of normal deviates in C order -- and its in-place accumulation in the array's own dtype.

What the sizes pin, and what they do not:

* `FlacArray.nbytes` is the size of the compressed blob (src/flacarray/array.py:135,197): per stream, libFLAC writes
  "fLaC" (4 B) + STREAMINFO (4 + 34 B) + a VORBIS_COMMENT holding only the vendor string (4 + 4 + 32 + 4 = 44 B:
  "reference libFLAC 1.4.3 20230623" and "reference libFLAC 1.5.0 20250211" are both 32 characters) = 86 B of stream
  header, then the frames.  This encoder writes "fLaC" + STREAMINFO + a SEEKTABLE of one 18-byte point per frame
  (4 + 18 F B) = 46 + 18 F B.  So   published - 86 S   =   total FRAME bytes libFLAC produced   (S streams),
  and the check is   ours - sum over streams (46 + 18 F)   ==   published - 86 S.
  All three arrays have 10 000-sample streams = 3 frames at level 5 (4096 + 4096 + 1808), i.e. 100 B of header here
  against 86 B there: ours == published + 14 S.
* Pinned: the total size of the frames libFLAC 1.4/1.5 writes at level 5 (`compress.py:12` default) for 14 streams /
  42 frames, 12 of the streams two-channel (int64 as low word + high word, utils.c:96-123): predictor choice, order,
  coefficient precision and quantisation, Rice partition order and parameters all have to land on the same bit counts,
  and the float quantisation (utils.c:160-243, :245-327; `precision` -> quanta = std / 10^p, utils.py:282-296) has to
  produce the same integers.  Levels 0-2 miss by 400-1600 B; level 3 misses the float32 cases by 12 and 20 B (on the
  float64 array -- 34-bit integers: low words VERBATIM, high words |v| <= 10 with LPC of order 3-4 -- levels 3-8 of
  this encoder write the same sizes), so the check discriminates.
* Not pinned: the order of the bytes inside the frames, levels other than 5, libFLAC reading these streams.
"""
import numpy as np

SEED = 123456789  # /root/reference/src/flacarray/demo.py:12

# (shape, dtype, keyword of FlacArray.from_array / array_compress, published nbytes, where)
SIZES = [
    ((10000,), np.float32, {"quanta": 1.0e-8}, 36161, "docs/docs/cookbook.ipynb cells 15, 18, 20"),
    ((10000,), np.float32, {"quanta": 1.0e-4}, 19664, "docs/docs/cookbook.ipynb cells 15, 28, 30"),
    ((4, 3, 10000), np.float64, {"precision": 10}, 522899, "docs/docs/tutorial.ipynb cells 4, 7, 8"),
]

LIBFLAC_STREAM_HEADER = 4 + (4 + 34) + (4 + 4 + 32 + 4)  # = 86


def own_stream_header(stream_size, level=5):
    """Bytes this encoder writes in front of a stream's first frame: fLaC + STREAMINFO + SEEKTABLE."""
    block = 1152 if level <= 2 else 4096
    frames = -(-stream_size // block)
    return 4 + (4 + 34) + (4 + 18 * frames)


def frame_bytes_published(case):
    shape, _, _, published, _ = case
    n_stream = int(np.prod(shape[:-1])) if len(shape) > 1 else 1
    return published - LIBFLAC_STREAM_HEADER * n_stream


# cookbook.ipynb cells 4, 5, 12: (1000, 100000) float32, quanta 1e-7, stream 500 restored.  The cell asks for the last
# 10 000 samples with slice(-n_end, None, 1), which the reference forwards as negative numbers = "no slice"
# (array.py:564-565, decompress.c:209): what it printed is the WHOLE stream, so the six visible
# numbers are samples 0, 1, 2 and 99 997, 99 998, 99 999.
VALUES_SHAPE = (1000, 100000)
VALUES_STREAM = 500
VALUES_QUANTA = 1.0e-7
VALUES_INDEX = (0, 1, 2, -3, -2, -1)
VALUES_PRINTED = ("1.3499217", "1.1607051", "-1.0080613", "0.2447555", "1.0821551", "0.03497732")
# What utils.c:350-368 gives on the integers of utils.c:232-240 for these samples, in this build and in the oracle:
# samples 0, 1 and 99 998 print as published; samples 2, 99 997 and 99 999 come out 2, 1 and 1 units in the last place
# of the restore PRODUCT coeff * (float)i lower (2.4e-7, 6.0e-8, 6.0e-8; the quanta is 1e-7 and float32 resolves
# 1.2e-7 at these magnitudes).  Neither a fused multiply-add nor an integer one step away reproduces all three, so
# the notebook's build or inputs differ from this restatement somewhere below float32 resolution.  The tests allow
# two units of the product.
VALUES_ULPS_OF_PRODUCT = 2


def _wave(n, dtype, sigma):
    """demo.py:77-82: two sines accumulated in an array of the OUTPUT dtype (float32 rounds after each term)."""
    t = np.arange(n)
    base = 5 / n
    acc = np.zeros(n, dtype=dtype)
    for mult, amp in ((3, 2 * sigma), (1, 6 * sigma)):
        acc += amp * np.sin(2 * np.pi * (mult * base) * t)
    return acc


def fake_data(shape, dtype, sigma=1.0, dc_sigma=5, seed=SEED):
    """`create_fake_data(shape, sigma, dtype, seed)[0]` for a float dtype and no communicator (demo.py:70-97,113-114)."""
    shape = tuple(shape)
    lead = shape[:-1]
    rng = np.random.default_rng(seed)
    dc = dc_sigma * sigma * (rng.random(size=lead + (1,)) - 0.5)
    wave = _wave(shape[-1], dtype, sigma)
    scale = rng.random(size=lead + (1,))
    out = np.empty(shape, dtype=dtype)
    out[...] = dc
    out += scale * wave
    out += rng.normal(0.0, sigma, int(np.prod(shape))).reshape(shape)
    if out.ndim == 2 and out.shape[0] == 1:
        out = out.reshape(-1)
    return out


def fake_data_stream(shape, dtype, index, sigma=1.0, dc_sigma=5, seed=SEED):
    """Row `index` of `fake_data(shape, dtype)` for a 2-D shape without materialising the array: the normal deviates
    of the rows before it are drawn (the generator's stream is sequential) and dropped."""
    n_row, n = shape
    rng = np.random.default_rng(seed)
    dc = dc_sigma * sigma * (rng.random(size=(n_row, 1)) - 0.5)
    wave = _wave(n, dtype, sigma)
    scale = rng.random(size=(n_row, 1))
    for _ in range(index):
        rng.normal(0.0, sigma, n)
    out = np.empty(n, dtype=dtype)
    out[...] = dc[index]
    out += scale[index] * wave
    out += rng.normal(0.0, sigma, n)
    return out
