"""BeamSearchDecoder (reference: nabu/neuralnetworks/decoders/beam_search_decoder.py:11-196 over
components/beam_search_decoder.py): beam search through the model's recurrent attention decoder.
The encoder runs once, then rnn_decoder.beam_search (nabu_speller_beam_search) does the search on
the device.  Evaluation = edit distance of the best beam against the reference without its
end-of-sequence label, divided by the reference lengths INCLUDING that label, as the reference
computes it (decoders/beam_search_decoder.py:171-183)."""
import os

import numpy as np
import torch

from nabu_amd import ops
from nabu_amd import variables as vs
from nabu_amd.autodiff import SeqLen
from nabu_amd.neuralnetworks.decoders import decoder
from nabu_amd.neuralnetworks.models.ed_decoders import rnn_decoder


class BeamSearchDecoder(decoder.Decoder):
    '''Beam search decoder; conf: max_steps, beam_width, length_penalty, temperature,
    visualize_alignments, alphabet (defaults/beamsearchdecoder.cfg)'''

    def __init__(self, conf, model):
        super(BeamSearchDecoder, self).__init__(conf, model)
        self.alphabet = self.conf['alphabet'].split(' ')

    def _keep_alignments(self):
        return self.conf.get('visualize_alignments') == 'True'

    def __call__(self, inputs, input_seq_length):
        '''Returns {output: (sequences [B,W,time], lengths [B,W], scores [B,W],
        alignments [B,W,time,Te] or None)}, beams best first'''
        model = self.model
        output_name = list(model.output_dims.keys())[0]
        with torch.no_grad(), vs.as_default(model.store):
            encoded, encoded_seq_length = model.encoder(inputs=inputs, input_seq_length=input_seq_length,
                                                        is_training=False)
            # the decoder's own scope, so the cell's variables are the trained ones
            with vs.variable_scope(model.decoder.scope):
                cell = model.decoder.create_cell(encoded, encoded_seq_length, False)
                e = list(encoded.keys())[0]
                res = rnn_decoder.beam_search(
                    cell, encoded[e], encoded_seq_length[e], beam_width=int(self.conf['beam_width']),
                    max_steps=int(self.conf['max_steps']), length_penalty=float(self.conf['length_penalty']),
                    temperature=float(self.conf['temperature']), with_alignments=self._keep_alignments())
        return {output_name: res}

    def write(self, outputs, directory, names):
        '''per utterance a file <name> with one "<score> <symbols>" line per beam, and the
        alignments of all beams as <name>_alignments.npy when visualize_alignments is set'''
        sequences, lengths, scores, alignments = list(outputs.values())[0]
        sequences, lengths, scores = sequences.cpu().numpy(), lengths.cpu().numpy(), scores.cpu().numpy()
        for i, name in enumerate(names):
            with open(os.path.join(directory, name), 'w') as fid:
                for b in range(sequences.shape[1]):
                    text = ' '.join(self.alphabet[s] for s in sequences[i, b, :lengths[i, b]])
                    fid.write('%f %s\n' % (scores[i, b], text))
            if alignments is not None:
                np.save(os.path.join(directory, name + '_alignments.npy'), alignments[i].cpu().numpy())

    def update_evaluation_loss(self, loss, outputs, references, reference_seq_length):
        sequences, lengths, _, _ = list(outputs.values())[0]
        dev = sequences.device
        best, best_len = sequences[:, 0].contiguous(), lengths[:, 0].contiguous()
        if best.shape[1] == 0:
            best = torch.zeros((best.shape[0], 1), dtype=torch.int32, device=dev)
        ref = torch.as_tensor(list(references.values())[0]).to(torch.int32).to(dev)
        ref_len = SeqLen.wrap(list(reference_seq_length.values())[0], dev)
        no_eos = (ref_len.dev - 1).contiguous()
        errors = int(ops.edit_distance(best, best_len, ref, no_eos).sum().item())
        self._fold(loss, errors, int(decoder.host_lengths(ref_len).sum()))
