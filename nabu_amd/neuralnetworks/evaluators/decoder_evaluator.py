"""DecoderEvaluator (reference: nabu/neuralnetworks/evaluators/decoder_evaluator.py:8-57): the
validation/test measure is the error rate of a decoder (CTC beam search or attention beam search)
run on the model — decoding and scoring both happen on the device."""
import torch

from nabu_amd.autodiff import SeqLen
from nabu_amd.neuralnetworks.decoders import decoder_factory
from nabu_amd.neuralnetworks.evaluators import evaluator


class DecoderEvaluator(evaluator.Evaluator):
    '''evaluates a decoder; conf needs a [decoder] section next to [evaluator]'''

    def __init__(self, conf, dataconf, model):
        super(DecoderEvaluator, self).__init__(conf, dataconf, model)
        self.decoder = decoder_factory.factory(conf.get('decoder', 'decoder'))(conf, model)

    def reset(self):
        self.decoder.reset()

    def update_loss(self, loss, batch):
        dev = torch.device('cuda', torch.cuda.current_device())
        inputs = {n: torch.as_tensor(a).to(torch.float32).to(dev) for n, a in batch['inputs'].items()}
        il = {n: SeqLen.wrap(a, dev) for n, a in batch['input_seq_length'].items()}
        outputs = self.decoder(inputs, il)
        self.decoder.update_evaluation_loss(loss, outputs, batch['targets'], batch['target_seq_length'])
