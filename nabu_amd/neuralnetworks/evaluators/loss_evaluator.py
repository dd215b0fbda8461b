"""LossEvaluator (reference: nabu/neuralnetworks/evaluators/loss_evaluator.py:8-64): the
validation loss is the utterance-weighted running mean of the training loss function
evaluated with is_training=False, on the HIP forward path (no tape, no backward)."""
import torch

from nabu_amd.autodiff import SeqLen
from nabu_amd.neuralnetworks.evaluators import evaluator
from nabu_amd.neuralnetworks.trainers import loss_functions


class LossEvaluator(evaluator.Evaluator):
    '''evaluates a loss function on the validation set'''

    def reset(self):
        self.num_utt = 0.0

    def update_loss(self, loss, batch):
        dev = torch.device('cuda', torch.cuda.current_device())
        def put(d, dt):
            return {n: torch.as_tensor(a).to(dt).to(dev) for n, a in d.items()}
        inputs, targets = put(batch['inputs'], torch.float32), put(batch['targets'], torch.int32)
        il = {n: SeqLen.wrap(a, dev) for n, a in batch['input_seq_length'].items()}
        tl = {n: SeqLen.wrap(a, dev) for n, a in batch['target_seq_length'].items()}
        with torch.no_grad():
            logits, logit_seq_length = self.model(inputs, il, targets, tl, False)
            batch_loss = loss_functions.factory(self.conf['loss'])(targets, logits, logit_seq_length, tl)
        loss_functions.check_status()
        batch_utt = float(list(logits.values())[0].shape[0])
        new_num_utt = self.num_utt + batch_utt
        loss[0] = (loss[0] * self.num_utt + float(batch_loss.item()) * batch_utt) / new_num_utt
        self.num_utt = new_num_utt
