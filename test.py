#!/usr/bin/env python
"""Evaluation CLI with the reference's flags (reference ``test.py:12-26``): greedy-decode a manifest with a trained
checkpoint and report corpus-level WER / CER (``test.py:81-104``: total edits / total reference words, chars)."""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, 'aes-lac-2018_amd'))

from codes.data import AudioDataLoader, AudioDataset  # noqa: E402
from codes.decoder import BeamCTCDecoder, GreedyDecoder  # noqa: E402
from codes.transforms import BatchSpectrogram, waveform_scale  # noqa: E402
from codes.utils.model_utils import load_model  # noqa: E402


def main(argv=None):
    p = argparse.ArgumentParser(description='DeepSpeech transcription')
    p.add_argument('--data-dir')
    p.add_argument('--model-path', default='models/deepspeech_final.pth')
    p.add_argument('--cuda', action='store_true', help='kept for compatibility: the model always runs on the GPU')
    p.add_argument('--manifest', metavar='DIR', default='data/test_manifest.csv')
    p.add_argument('--batch-size', default=32, type=int)
    p.add_argument('--num-workers', default=4, type=int)
    p.add_argument('--decoder', default='greedy', choices=['greedy', 'beam', 'none'], type=str,
                   help="'beam' (CTC prefix beam search, no LM) is an addition to the reference's greedy / none")
    p.add_argument('--beam-width', default=16, type=int)
    p.add_argument('--verbose', action='store_true')
    p.add_argument('--output-path', default=None, type=str)
    args = p.parse_args(argv)

    torch.set_grad_enabled(False)
    model, _, val_t, target_t = load_model(args.model_path, return_transforms=True, data_dir=args.data_dir)
    model.eval().to('cuda')
    target_t = target_t[0]
    decoder = {'greedy': lambda: GreedyDecoder(target_t.label_encoder),
               'beam': lambda: BeamCTCDecoder(target_t.label_encoder, beam_width=args.beam_width),
               'none': lambda: None}[args.decoder]()
    target_decoder = GreedyDecoder(target_t.label_encoder)
    dataset = AudioDataset(args.data_dir, args.manifest, transforms=val_t, target_transforms=target_t)
    loader = AudioDataLoader(dataset, batch_size=args.batch_size, num_workers=args.num_workers, raw_audio=True)
    frontend = BatchSpectrogram(device='cuda', scale=waveform_scale(val_t))

    total_cer = total_wer = num_tokens = num_chars = 0
    output_data = []
    for wavs, targets, _, target_sizes in loader:
        inputs, input_percentages = frontend(wavs)
        out = model(inputs)                                                     # (B,T,A) probabilities
        sizes = input_percentages.mul_(int(out.shape[1])).int()                 # test.py:70-71
        if decoder is None:
            output_data.append((out.cpu().numpy(), sizes.numpy()))
            continue
        decoded, _ = decoder.decode(out, sizes)
        off = 0
        for i, n in enumerate(target_sizes.tolist()):
            reference = target_decoder.convert_to_strings([targets[off:off + n]])[0][0]
            off += n
            transcript = decoded[i][0]
            w, c = decoder.wer(transcript, reference), decoder.cer(transcript, reference)
            total_wer += w
            total_cer += c
            num_tokens += len(reference.split())
            num_chars += len(reference)
            if args.verbose:
                print('Ref: {}\nHyp: {}\nWER: {}\t CER: {}\n'.format(reference.lower(), transcript.lower(),
                                                                      w / max(1, len(reference.split())),
                                                                      c / max(1, len(reference))))
    if decoder is not None:
        print('Test Summary \tAverage WER {wer:.3f}\tAverage CER {cer:.3f}\t'.format(
            wer=100.0 * total_wer / max(1, num_tokens), cer=100.0 * total_cer / max(1, num_chars)))
    else:
        np.save(args.output_path, np.asarray(output_data, dtype=object), allow_pickle=True)


if __name__ == '__main__':
    main()
