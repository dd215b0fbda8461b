"""Neural network trainer environment (reference:
nabu/neuralnetworks/trainers/trainer.py:18-792).

Same constructor arguments, same ``[trainer]`` cfg keys and defaults, same training
loop (read a batch -> update -> print -> increment the global step).  What changed
underneath:
  * the TF graph of one step is a sequence of HIP kernel launches on one stream
    (model forward, fused loss+gradient, hand-written backward kernels);
  * ``tf.train.AdamOptimizer`` + per-variable ``clip_by_value`` (trainer.py:525,
    560-569) is ONE fused clip+Adam kernel over a flat parameter buffer;
  * the parameter-server data parallelism (trainer.py:479-510, 822-914) is one
    RCCL all-reduce of the flat gradient buffer per step: every replica clips its
    own gradient, the clipped gradients are averaged, one Adam update is applied —
    the synchronous semantics the reference's SyncReplicasOptimizer branch
    intends with numbatches_to_aggregate = number of workers."""
import math
import os
import pickle
import time
from abc import ABCMeta, abstractmethod

import numpy as np
import torch

from nabu_amd import ops as hip
from nabu_amd.autodiff import Tape, SeqLen
from nabu_amd.neuralnetworks.models.model import Model
from nabu_amd.neuralnetworks.trainers import loss_functions
from nabu_amd.tools.default_conf import apply_defaults, defaults_path

ADAM_B1, ADAM_B2, ADAM_EPS = 0.9, 0.999, 1e-8      # tf.train.AdamOptimizer defaults
CLIP = 1.0                                         # tf.clip_by_value(grad, -1., 1.)


class Trainer(object, metaclass=ABCMeta):
    '''General class outlining the training environment of a model.'''

    def __init__(self, conf, dataconf, modelconf, evaluatorconf, expdir, server, task_index):
        '''
        Args:
            conf: the trainer config as a ConfigParser ([trainer] section)
            dataconf: the data source: an object with ``batch(step)`` and
                ``num_batches()`` (e.g. processing.synthetic.SyntheticData), or a
                ConfigParser with a [synthetic] section describing one
            modelconf: the model configuration (ConfigParser: io/encoder/decoder)
            evaluatorconf: the evaluator configuration, None or evaluator = None
                disables validation
            expdir: directory where the model is written
            server: data-parallel process group (computing.dist.create_server())
                or None — replaces the tf.train.Server of the reference
            task_index: index of this worker (rank)
        '''
        self.conf = dict(conf.items('trainer'))
        apply_defaults(self.conf, defaults_path(__file__, self))
        self.dataconf = dataconf
        self.evaluatorconf = evaluatorconf
        self.expdir = expdir
        self.server = server
        self.task_index = task_index
        if 'norm_constraint' in self.conf and self.conf['norm_constraint'] != 'None':
            raise Exception('norm_constraint (MaxNorm) is off by default in the reference '
                            '(standardtrainer.cfg:18) and not on the MI355X hot path')
        if int(self.conf['cut_sequence_length']):
            raise Exception('cut_sequence_length is part of the input pipeline (SURVEY.md 8(f) '
                            'row 3) and not supported yet')
        self.model = Model(conf=modelconf, trainlabels=int(self.conf['trainlabels']), constraint=None)
        self._graph = None

    # ------------------------------------------------------------------ graph
    def _create_graph(self):
        '''set up everything one step needs (the analogue of building the TF graph,
        reference trainer.py:73-283)'''
        if self._graph is not None:
            return self._graph
        outputs = {}
        self.global_step = 0
        self.adam_step = 0
        self.learning_rate_fact = 1.0
        self.data = self._data()
        outputs['num_steps'] = self.data.num_batches() * int(self.conf['num_epochs'])
        self.loss_fn = loss_functions.factory(self.conf['loss'])
        self.world = self.server.world_size if self.server is not None else 1
        self.flat = self.flat_grad = self.adam_m = self.adam_v = None
        self._graph = outputs
        return outputs

    def _data(self):
        '''the batch source (reference trainer.py:285-423 builds the queue-runner
        pipeline here)'''
        d = self.dataconf
        if hasattr(d, 'batch') and hasattr(d, 'num_batches'):
            return d
        if hasattr(d, 'has_section') and d.has_section('synthetic'):
            from nabu_amd.processing.synthetic import SyntheticData
            kw = {k: (v == 'True' if v in ('True', 'False') else int(v)) for k, v in d.items('synthetic')}
            kw.setdefault('batch_size', int(self.conf['batch_size']))
            return SyntheticData(**kw)
        raise Exception('dataconf must be a batch source or contain a [synthetic] section; the '
                        'TFRecord input pipeline is outside the hot path (SURVEY.md 2.1 row 9)')

    def learning_rate(self):
        '''exponential decay (non-staircase) times the validation factor
        (reference trainer.py:153-166)'''
        num_steps = self._graph['num_steps']
        return (float(self.conf['initial_learning_rate'])
                * float(self.conf['learning_rate_decay']) ** (float(self.global_step) / num_steps)
                * self.learning_rate_fact)

    def to_device(self, batch, device=None):
        '''numpy batch (A0 contract) -> device tensors / SeqLen objects'''
        if device is None:
            device = torch.device('cuda', torch.cuda.current_device())
        out = {}
        for key in ('inputs', 'targets'):
            out[key] = {}
            for n, a in batch[key].items():
                if isinstance(a, torch.Tensor):
                    out[key][n] = a.to(device)
                else:
                    dt = torch.float32 if key == 'inputs' else torch.int32
                    out[key][n] = torch.from_numpy(np.ascontiguousarray(a)).to(dt).to(device)
        for key in ('input_seq_length', 'target_seq_length'):
            out[key] = {n: SeqLen.wrap(a, device) for n, a in batch[key].items()}
        return out

    # ------------------------------------------------------------------- step
    def step(self, batch):
        '''one training update on a device batch; returns the loss as a 1-element
        device tensor (no host synchronisation)'''
        self._create_graph()
        with Tape() as tape:
            logits, logit_seq_length = self.model(
                inputs=batch['inputs'], input_seq_length=batch['input_seq_length'],
                targets=batch['targets'], target_seq_length=batch['target_seq_length'],
                is_training=True)
            loss = self.loss_fn(batch['targets'], logits, logit_seq_length, batch['target_seq_length'])
            extra = self.aditional_loss()
            if extra is not None:
                loss = hip.axpy_(loss, extra)
        if self.flat is None:
            self._init_optimizer()
        tape.backward(loss)
        self._update()
        return loss

    def _init_optimizer(self):
        store = self.model.store
        self.flat, self.flat_grad = store.flatten()
        self.adam_m = torch.zeros_like(self.flat)
        self.adam_v = torch.zeros_like(self.flat)
        if self.world > 1:
            self.server.broadcast_(self.flat, 0)       # identical replicas

    def _update(self):
        '''clip + Adam (reference trainer.py:512-580), with the gradient exchange of
        the data-parallel mode in between'''
        lr = self.learning_rate()
        self.adam_step += 1
        t = self.adam_step
        lr_t = lr * math.sqrt(1.0 - ADAM_B2 ** t) / (1.0 - ADAM_B1 ** t)
        if self.world > 1:
            hip.clip_(self.flat_grad, CLIP)                      # clip per replica ...
            self.server.all_reduce_sum_(self.flat_grad)          # ... sum over xGMI ...
            hip.adam_clip_step(self.flat, self.flat_grad, self.adam_m, self.adam_v, lr_t,
                               ADAM_B1, ADAM_B2, ADAM_EPS, CLIP, 1.0 / self.world)   # ... mean, Adam
        else:
            hip.adam_clip_step(self.flat, self.flat_grad, self.adam_m, self.adam_v, lr_t,
                               ADAM_B1, ADAM_B2, ADAM_EPS, CLIP, 1.0)
        self.last_lr = lr

    # ------------------------------------------------------------------ train
    def train(self, testing=False):
        '''train the model (reference trainer.py:582-792)

        args:
            testing: if true only the "graph" is created, for debugging purposes
        returns: the list of (global_step, loss, learning_rate) that was printed'''
        outputs = self._create_graph()
        if testing:
            return []
        is_chief = self.task_index == 0
        history = []
        num_steps = outputs['num_steps']
        while self.global_step < num_steps:
            start = time.time()
            # each replica reads its own shard of the epoch (replaces the shared
            # filename queue on ps:0, trainer.py:342-351)
            batch = self.to_device(self.data.batch(self.global_step * self.world + self.task_index))
            loss = self.step(batch)
            loss_functions.check_status()
            if self.world > 1:
                loss = self.server.all_reduce_sum_(loss.clone())
                loss_value = float(loss.item()) / self.world
            else:
                loss_value = float(loss.item())
            print(('WORKER %d: step %d/%d loss: %f, learning rate: %f \n\t time elapsed: %f sec'
                   '\n\t peak memory usage: %d/%d MB')
                  % (self.task_index, self.global_step, num_steps, loss_value, self.last_lr,
                     time.time() - start, torch.cuda.max_memory_allocated() / 1e6,
                     torch.cuda.get_device_properties(0).total_memory / 1e6))
            history.append((self.global_step, loss_value, self.last_lr))
            self.global_step += 1
        if is_chief and self.expdir is not None:
            self.save()
        return history

    def save(self):
        '''final model: variables by TF-style name (SaveAtEnd, hooks.py:30-52) and
        the pickled model configuration (trainer.py:789-792)'''
        mdir = os.path.join(self.expdir, 'model')
        os.makedirs(mdir, exist_ok=True)
        np.savez(os.path.join(mdir, 'network.ckpt.npz'), **self.model.store.state_dict())
        with open(os.path.join(mdir, 'model.pkl'), 'wb') as fid:
            pickle.dump({s: dict(self.model.conf.items(s)) for s in self.model.conf.sections()}, fid)

    @abstractmethod
    def aditional_loss(self):
        '''an additional loss term or None'''

    @abstractmethod
    def chief_only_hooks(self, outputs):
        '''hooks for the chief worker only'''

    @abstractmethod
    def hooks(self, outputs):
        '''hooks for every worker'''
