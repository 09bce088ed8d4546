"""Oracle: waveform decode and the training-set augmentation (tempo + gain) on the host, in numpy.

TEST INFRASTRUCTURE ONLY -- never imported by the product path (which does all of this on the GPU:
aes-lac-2018_amd/csrc/audio.hip).

What it restates (reference ``codes/transforms.py:130-224``, ``ToTensor``):
  * no augmentation: the clip as float samples (``torchaudio.load``; here 16-bit PCM / 32768, the scale the
    spectrogram's log1p was trained with is a property of the torchaudio version -- SURVEY.md 8c (3));
  * augmentation: ``sox <in> -r 16000 -c 1 -b 16 -e si <out> tempo T gain G`` with T, G drawn uniformly and printed
    with three decimals (``:173-201``).  sox is not in this image and its sources are not in the reference tree, so
    ``tempo`` is restated as the published WSOLA algorithm with sox's default parameters (82 ms segment, 14.68 ms
    search, 12 ms overlap) -- **parity unpinned** w.r.t. the sox binary: same kind of signal, not the same samples --
    followed by the gain in dB and the clip + rounding that writing ``-b 16`` implies.

The WSOLA below is the SPECIFICATION the device kernel is tested against bit for bit:
  * segment search: among the candidate starts ``lo..hi`` the one whose first ``ovl`` samples correlate best with the
    natural continuation of the previous segment; correlations are accumulated in float64 in sample order (products of
    two float32 values are exact in float64, so fused or separate multiply-add give the same bits); ties -> the
    lowest start;
  * cross-fade ``out = tail * (1 - f) + head * f`` in float32 with each product and the sum rounded (no FMA),
    ``f = float32(j * (1 / ovl))``;
  * the number of segments, hence the output length, depends only on (len(x), tempo) -- ``wsola_out_len``.
"""
import numpy as np

SEGMENT_MS, SEARCH_MS, OVERLAP_MS = 82.0, 14.68, 12.0


UNIT_SCALE = 1.0 / 32768.0          # samples in [-1, 1)
INT32_SCALE = 65536.0               # the un-normalised int32-range floats of the mid-2018 torchaudio master


def pcm16_to_float(pcm, scale=UNIT_SCALE):
    """int16 samples -> float32: q * scale, one rounding (none for a power-of-two scale).  The default gives [-1, 1)."""
    return np.asarray(pcm, dtype=np.int16).astype(np.float32) * np.float32(scale)


def gain_requantize(y, gain_db, scale=UNIT_SCALE):
    """``gain G`` then the 16-bit signed output format: y * 10^(G/20) (y in [-1, 1)), round to the nearest int16 step (half
    to even), clip -> an integer q; what a loader with amplitude scale ``scale`` then hands on is q * scale.  ``gain_db`` is
    first printed with three decimals, as the sox command line was."""
    g = np.float32(10.0 ** (float('{:.3f}'.format(gain_db)) / 20.0))
    y = np.asarray(y, dtype=np.float32) * g
    return np.clip(np.round(y * np.float32(32768.0)), -32768, 32767).astype(np.float32) * np.float32(scale)


def wsola_params(sample_rate=16000):
    seg = max(int(sample_rate * SEGMENT_MS / 1000.0), 4)
    ovl = max(min(int(sample_rate * OVERLAP_MS / 1000.0), seg // 2), 1)
    half = max(int(sample_rate * SEARCH_MS / 1000.0) // 2, 1)
    return seg, ovl, half


def wsola_plan(n, tempo, sample_rate=16000):
    """The data-INDEPENDENT part of the algorithm: list of (lo, hi) candidate ranges per segment, and the output length."""
    seg, ovl, half = wsola_params(sample_rate)
    tempo = float('{:.3f}'.format(tempo))
    if abs(tempo - 1.0) < 1e-6 or n == 0 or n <= seg + half:
        return None, n
    hop_out = seg - ovl
    hop_in = tempo * hop_out
    ranges, ideal, out_pos = [], 0.0, hop_out
    while True:
        ideal += hop_in
        base = int(round(ideal))
        lo, hi = max(base - half, 0), min(base + half, n - seg)
        if hi < lo:
            break
        ranges.append((lo, hi))
        out_pos += hop_out
    return ranges, out_pos + ovl


def wsola_out_len(n, tempo, sample_rate=16000):
    return wsola_plan(n, tempo, sample_rate)[1]


def wsola_tempo(x, tempo, sample_rate=16000):
    """Change the tempo of a 1-D float32 array without changing its pitch; output length ``wsola_out_len``."""
    x = np.asarray(x, dtype=np.float32).reshape(-1)
    seg, ovl, half = wsola_params(sample_rate)
    ranges, out_len = wsola_plan(x.size, tempo, sample_rate)
    if ranges is None:
        return x.copy()
    hop_out = seg - ovl
    fade = (np.arange(ovl, dtype=np.float64) * (1.0 / ovl)).astype(np.float32)
    one_minus = np.float32(1.0) - fade
    out = np.zeros(out_len, dtype=np.float32)
    out[:seg] = x[:seg]
    out_pos, prev = hop_out, 0
    x64 = x.astype(np.float64)
    for lo, hi in ranges:
        want = x[prev + hop_out: prev + hop_out + ovl]
        nd = hi - lo + 1
        corr = np.zeros(nd, dtype=np.float64)
        for j in range(ovl):                                   # sample order, float64: the kernel's order
            corr += x64[lo + j: lo + j + nd] * np.float64(want[j])
        start = lo + int(np.argmax(corr))                      # first maximum
        head = x[start: start + ovl]
        out[out_pos: out_pos + ovl] = want * one_minus + head * fade        # each product and the sum rounded to f32
        out[out_pos + ovl: out_pos + seg] = x[start + ovl: start + seg]
        prev, out_pos = start, out_pos + hop_out
    return out[:out_pos + ovl]


def augment(pcm, tempo, gain_db, sample_rate=16000, scale=UNIT_SCALE):
    """What a training clip goes through with ``augment=True``: decode, tempo, gain, 16-bit requantisation."""
    return gain_requantize(wsola_tempo(pcm16_to_float(pcm), tempo, sample_rate), gain_db, scale)
