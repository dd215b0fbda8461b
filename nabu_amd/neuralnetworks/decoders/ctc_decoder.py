"""CTCDecoder (reference: nabu/neuralnetworks/decoders/ctc_decoder.py:10-135): the model's
logits go through tf.nn.ctc_beam_search_decoder (beam 100, best path, merge_repeated) — here
nabu_ctc_beam_search, one workgroup per utterance on the device — and the error measure is the
label edit distance summed over the batch divided by the number of reference labels."""
import os

import torch

from nabu_amd import ops
from nabu_amd.autodiff import SeqLen
from nabu_amd.neuralnetworks.decoders import decoder

BEAM_WIDTH = 100          # tf.nn.ctc_beam_search_decoder's default, which the reference relies on


class CTCDecoder(decoder.Decoder):
    '''CTC Decoder; conf: <output>_alphabet = space separated symbols for every model output'''

    def __init__(self, conf, model):
        super(CTCDecoder, self).__init__(conf, model)
        self.alphabets = {o: self.conf['%s_alphabet' % o].split(' ') for o in model.output_names}

    def __call__(self, inputs, input_seq_length):
        '''Returns {output: (ids [B,T'] int32 padded with -1, lengths [B] int32)} — the dense form
        of the reference's SparseTensor'''
        with torch.no_grad():
            logits, logit_len = self.model(inputs, input_seq_length, targets=[], target_seq_length=[],
                                           is_training=False)
            outputs = {}
            for o in logits:
                lens = SeqLen.wrap(logit_len[o], logits[o].device)
                ids, out_len, _ = ops.ctc_beam_search(logits[o], lens.dev, BEAM_WIDTH, merge_repeated=True)
                outputs[o] = (ids, out_len)
        return outputs

    def write(self, outputs, directory, names):
        '''one line "<name> <symbols>" per utterance appended to <directory>/<output>'''
        for o, (ids, lens) in outputs.items():
            ids, lens = ids.cpu().numpy(), lens.cpu().numpy()
            with open(os.path.join(directory, o), 'a') as fid:
                for i, name in enumerate(names):
                    text = ' '.join(self.alphabets[o][j] for j in ids[i, :lens[i]])
                    fid.write('%s %s\n' % (name, text))

    def update_evaluation_loss(self, loss, outputs, references, reference_seq_length):
        '''label error rate: sum of edit distances / number of reference labels, running over
        the batches seen since reset()'''
        errors, batch_targets = 0, 0
        for o, (ids, lens) in outputs.items():
            dev = ids.device
            ref = torch.as_tensor(references[o]).to(torch.int32).to(dev)
            ref_len = SeqLen.wrap(reference_seq_length[o], dev)
            errors += int(ops.edit_distance(ids, lens, ref, ref_len.dev).sum().item())
        for lengths in reference_seq_length.values():
            batch_targets += int(decoder.host_lengths(lengths).sum())
        self._fold(loss, errors, batch_targets)
