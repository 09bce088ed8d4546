"""CPU oracle for the DeepSpeech2 hot path -- TEST INFRASTRUCTURE ONLY.

This package restates, on the CPU, the arithmetic of the reference's
spectrogram -> conv -> BiGRU -> CTC path (igormq/aes-lac-2018).  It exists to
check the HIP product path; nothing under ``aes-lac-2018_amd/`` may import it.
Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg use it.

Pinning status (see DESIGN.md "Oracle"):
  * model (conv/BN/GRU/FC): pinned against the reference's own
    ``codes/model.py`` imported in the build container; golden logits, probs and
    gradients are committed under ``tests/golden/`` (made by
    ``tests/golden/make_golden.py``).
  * CTC: the reference delegates to warp-ctc (SeanNaren fork, un-pinned HEAD,
    absent from the reference tree); restated from the published CTC
    forward/backward algorithm and pinned by a float64 brute-force path
    enumeration and by ``torch.nn.functional.ctc_loss`` -- "parity unpinned"
    with respect to warp-ctc itself.
  * spectrogram: the reference delegates to librosa (un-pinned, absent);
    restated from ``codes/transforms.py:94-119`` and cross-checked against
    ``torch.stft`` -- "parity unpinned" with respect to librosa itself.
"""
