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
    intends with numbatches_to_aggregate = number of workers.  Optional
    (``allreduce_buckets = True``, an extension key; default False): the same sum
    exchanged as one bucket per encoder layer (+ one for the decoder), each started
    as soon as its gradients are final and overlapped with the dense products of the
    next layer's backward pass — never with a persistent recurrent kernel."""
import math
import sys
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
CHECKPOINT_SECS = 600                              # MonitoredTrainingSession default


class ValidationSaveHook(object):
    '''saves and restores the validated model (reference components/hooks.py:54-86: a
    tf.train.Saver over ALL global variables, i.e. weights, Adam slots, global step,
    learning-rate factor and the validation bookkeeping).  Kept in host memory when the
    trainer has no expdir.'''

    def __init__(self, filename, trainer):
        self.filename = filename
        self.trainer = trainer
        self._mem = None

    def save(self):
        st = self.trainer.state()
        if self.filename is None or self.trainer.task_index != 0:
            self._mem = st
        else:
            os.makedirs(os.path.dirname(self.filename), exist_ok=True)
            torch.save(st, self.filename)
            self._mem = st

    def restore(self):
        if self._mem is None and self.filename is not None and os.path.exists(self.filename):
            self._mem = torch.load(self.filename, weights_only=False)
        if self._mem is None:
            raise Exception('no validated model to restore')
        self.trainer.load_state(self._mem)


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
            # reference trainer.py:364-384, 917-973: needs frame-synchronous targets (assert_equal over all
            # sequence lengths) and cannot build its graph on the pinned stack (tf.ceil over an int32
            # floor division, trainer.py:929) -- a non-zero key raises there too
            raise Exception('cut_sequence_length is not supported (it requires inputs and targets of equal '
                            'lengths and fails at graph construction in the reference, DESIGN.md section 8)')
        self.model = Model(conf=modelconf, trainlabels=int(self.conf['trainlabels']), constraint=None)
        if bool(getattr(server, 'shared_devices', False)):
            # ranks that share a GPU (more ranks than devices on this node: first-contact runs, tests): the persistent
            # recurrent kernels need every CU of the device for their own workgroups, two processes' launches would each
            # hold a part of it and spin into the bounded time-out — decided HERE, where the process group is known, for
            # every layer this process builds (training and validation alike)
            from nabu_amd import ops as _hip_ops
            from nabu_amd.neuralnetworks.components import layer as _layer
            _layer.LSTM_MODE[0] = _hip_ops.LSTM_STEPWISE
        self._graph = None
        self.schedule_log = None       # a list here records the bucketed exchange: (bucket key, 'hook' | 'final')

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
        self.buckets = None
        # validation part (reference trainer.py:189-265): bookkeeping 'variables' of the
        # validate/ scope with the reference's initial values
        self.evaluator = None
        if (self.evaluatorconf is not None and self.evaluatorconf.has_section('evaluator')
                and self.evaluatorconf.get('evaluator', 'evaluator') != 'None'):
            self.evaluator = self._validate()
        self.validated_step = -int(self.conf['valid_frequency'])
        self.best_validation = 1.79e+308
        self.num_tries = 0
        self.validation_hook = ValidationSaveHook(
            None if self.expdir is None else os.path.join(self.expdir, 'logdir', 'validated.ckpt'), self)
        self._graph = outputs
        return outputs

    def _validate(self):
        '''create the evaluator (reference trainer.py:459-477)'''
        from nabu_amd.neuralnetworks.evaluators import evaluator_factory
        evaltype = self.evaluatorconf.get('evaluator', 'evaluator')
        # a synthetic batch source derives its own validation set; on-disk data are described by
        # the database conf (dev sections named in the evaluator conf)
        dataconf = self.data if hasattr(self.data, 'validation') else self.dataconf
        return evaluator_factory.factory(evaltype)(conf=self.evaluatorconf, dataconf=dataconf, model=self.model)

    def validation_loss(self):
        '''run the evaluator over the validation set (reference trainer.py:660-680)'''
        loss, update_loss, valbatches = self.evaluator.evaluate()
        for i in range(valbatches):
            update_loss(i)
        return loss[0]

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
        # the reference's on-disk data (reference trainer.py:289-340): the trainer conf links the model's
        # input names and the target names to sections of database.conf
        from nabu_amd.processing import input_pipeline
        input_names = [n for n in self.model.conf.get('io', 'inputs').split(' ') if n]
        target_names = [n for n in self.conf['targets'].split(' ') if n]
        world = self.server.world_size if self.server is not None else 1
        return input_pipeline.from_sections(
            d, input_names, [self.conf[i].split(' ') for i in input_names],
            target_names, [self.conf[o].split(' ') for o in target_names],
            batch_size=int(self.conf['batch_size']), numbuckets=int(self.conf['numbuckets']),
            variable_batch_size=self.conf['variable_batch_size'] == 'True', shuffle=True, seed=0)

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
        self._backward_and_update(tape, loss)
        return loss

    def _backward_and_update(self, tape, loss):
        if self.buckets is not None:
            self._begin_bucketed(tape)
        try:
            tape.backward(loss)
        finally:
            if self.buckets is not None:
                self._end_backward()
                failed = hip.take_phase_hook_error()      # raised inside the C -> Python hook of nabu_blstm_bwd
                if failed is not None:
                    self._join_comm()                     # no collective outlives the step
                    raise failed
        self._update()

    def _init_optimizer(self):
        store = self.model.store
        self.flat, self.flat_grad = store.flatten()
        self.adam_m = torch.zeros_like(self.flat)
        self.adam_v = torch.zeros_like(self.flat)
        if self.world > 1:
            self.server.broadcast_(self.flat, 0)       # identical replicas
        self.buckets = None
        if self.world > 1 and self.conf.get('allreduce_buckets', 'False') == 'True':
            self.buckets = self._make_buckets()

    # ----------------------------------------------------- bucketed gradient exchange
    @staticmethod
    def bucket_key(name):
        '''variables of one encoder layer share a bucket (<encoder>/<input>/layer<l>), everything
        else (the decoder) forms one more'''
        parts = name.split('/')
        return '/'.join(parts[:3]) if len(parts) > 3 and parts[2].startswith('layer') else 'decoder'

    def _make_buckets(self):
        '''contiguous ranges of the flat gradient buffer, in variable-creation order'''
        buckets = []
        for v in self.model.store.trainable_variables():
            key = self.bucket_key(v.name)
            end = v.offset + (v.numel() + 3) // 4 * 4
            if buckets and buckets[-1]['key'] == key and buckets[-1]['end'] == v.offset:
                buckets[-1]['end'] = end
                buckets[-1]['vars'].append(v)
            else:
                buckets.append(dict(key=key, start=v.offset, end=end, vars=[v]))
        assert buckets[0]['start'] == 0 and buckets[-1]['end'] == self.flat_grad.numel()
        return buckets

    def _begin_bucketed(self, tape):
        self._ready = set()            # ids of variables whose gradient is final
        self._sent = set()             # bucket indices already on the wire
        self._works = []
        self._declared = {id(v) for op in tape.ops for v in op.params}
        tape.on_param_ready = lambda v: self._ready.add(id(v))
        # weight-gradient products deferred behind the last recurrence (layer.py, Tape.defer): a layer's bucket goes on
        # the wire right after its products and travels under the next layer's
        tape.after_deferred = self._launch_ready_buckets
        hip.set_phase_hook(self._launch_ready_buckets)
        hip.BEFORE_RECURRENT[0] = self._join_comm

    def _end_backward(self):
        hip.set_phase_hook(None)
        hip.BEFORE_RECURRENT[0] = None
        if sys.exc_info()[0] is not None:                 # the backward pass raised: drain what is in flight
            try:
                self._join_comm()
            except Exception:                             # the original error is the one to report
                pass

    def _launch_ready_buckets(self, everything=False):
        '''clip (per replica, before the aggregation: reference trainer.py:556-569) and start the
        all-reduce of every bucket whose gradients are final.  Called between a recurrent kernel and
        the dense products that follow it (nabu_blstm_set_phase_hook) and once after the backward pass.'''
        for i, b in enumerate(self.buckets):
            if i in self._sent:
                continue
            if not everything and not all(id(v) in self._ready and id(v) in self._declared for v in b['vars']):
                continue
            view = self.flat_grad[b['start']:b['end']]
            hip.clip_(view, CLIP)
            self._works.append(self.server.all_reduce_sum_async(view))
            self._sent.add(i)
            if self.schedule_log is not None:       # (bench.py / tests: which bucket went on the wire at which call)
                self.schedule_log.append((b['key'], 'final' if everything else 'hook'))

    def _join_comm(self):
        '''the launch stream waits for every exchange in flight (before a recurrent launch and
        before the optimiser)'''
        works, self._works = self._works, []
        for w in works:
            if w is not None:
                w.wait()

    def _update(self):
        '''clip + Adam (reference trainer.py:512-580), with the gradient exchange of
        the data-parallel mode in between'''
        lr = self.learning_rate()
        self.adam_step += 1
        t = self.adam_step
        lr_t = lr * math.sqrt(1.0 - ADAM_B2 ** t) / (1.0 - ADAM_B1 ** t)
        if self.world > 1 and self.buckets is not None:
            ev = self._allreduce_events()                        # exposed part of the exchange only
            self._launch_ready_buckets(everything=True)
            self._join_comm()
            if ev is not None:
                ev[1].record()
            hip.adam_clip_step(self.flat, self.flat_grad, self.adam_m, self.adam_v, lr_t,
                               ADAM_B1, ADAM_B2, ADAM_EPS, CLIP, 1.0 / self.world)
        elif self.world > 1:
            hip.clip_(self.flat_grad, CLIP)                      # clip per replica ...
            ev = self._allreduce_events()
            self.server.all_reduce_sum_(self.flat_grad)          # ... sum over xGMI ...
            if ev is not None:
                ev[1].record()
            hip.adam_clip_step(self.flat, self.flat_grad, self.adam_m, self.adam_v, lr_t,
                               ADAM_B1, ADAM_B2, ADAM_EPS, CLIP, 1.0 / self.world)   # ... mean, Adam
        else:
            hip.adam_clip_step(self.flat, self.flat_grad, self.adam_m, self.adam_v, lr_t,
                               ADAM_B1, ADAM_B2, ADAM_EPS, CLIP, 1.0)
        self.last_lr = lr

    def _allreduce_events(self):
        '''bench.py sets ``time_allreduce``: a pair of events on the launch stream around the
        exchange (the collective's own stream is joined to it by torch.distributed)'''
        if not getattr(self, 'time_allreduce', False) or not torch.cuda.is_available():
            return None
        ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
        ev[0].record()
        if not hasattr(self, 'allreduce_ms'):
            self.allreduce_ms = []
        self.allreduce_ms.append(ev)
        return ev

    # ------------------------------------------------------- state (checkpoints)
    def state(self):
        '''everything tf.train.Saver() of the reference's session would write: model
        variables, Adam slots, global step, learning-rate factor, validation bookkeeping'''
        if self.flat is None:
            self._init_optimizer()
        from nabu_amd.neuralnetworks.components import ops as nops
        return dict(names=list(self.model.store.order), flat=self.flat.detach().cpu().clone(),
                    adam_m=self.adam_m.cpu().clone(), adam_v=self.adam_v.cpu().clone(),
                    global_step=self.global_step, adam_step=self.adam_step,
                    learning_rate_fact=self.learning_rate_fact, validated_step=self.validated_step,
                    best_validation=self.best_validation, rng_offset=nops.global_rng().offset)

    def load_state(self, st):
        if self.flat is None:
            self._init_optimizer()
        if list(st['names']) != list(self.model.store.order) or st['flat'].numel() != self.flat.numel():
            raise Exception('checkpoint does not match the model (variable names / sizes differ)')
        from nabu_amd.neuralnetworks.components import ops as nops
        self.flat.copy_(st['flat'])
        self.adam_m.copy_(st['adam_m'])
        self.adam_v.copy_(st['adam_v'])
        self.global_step, self.adam_step = int(st['global_step']), int(st['adam_step'])
        self.learning_rate_fact = float(st['learning_rate_fact'])
        self.validated_step = int(st['validated_step'])
        self.best_validation = float(st['best_validation'])
        nops.global_rng().offset = int(st['rng_offset'])

    def _ensure_variables(self):
        '''variables are created by the first call of the model (the reference creates them
        when it builds the graph): run one forward pass if that has not happened yet'''
        if self.flat is None and not self.model.store.order:
            b = self.to_device(self.data.batch(0))
            with torch.no_grad():
                self.model(b['inputs'], b['input_seq_length'], b['targets'], b['target_seq_length'], False)

    def _validation_point(self):
        '''the validation branch of the training loop (reference trainer.py:646-737).
        Every replica evaluates the same validation set on identical weights, so all of them
        take the same decision (the reference lets the chief decide and the others wait).
        Returns True when training must terminate.'''
        print('WORKER %d: validating model' % self.task_index)
        prev_val_loss = self.best_validation
        validation_loss = self.validation_loss()
        print('WORKER %d: validation loss: %f' % (self.task_index, validation_loss))
        self.validation_history.append((self.global_step, validation_loss))
        if validation_loss >= prev_val_loss:
            print('WORKER %d: validation loss is worse' % self.task_index)
            if self.conf['num_tries'] != 'None':
                if self.num_tries == int(self.conf['num_tries']):
                    self.validation_hook.restore()
                    print('WORKER %d: terminating training' % self.task_index)
                    return True
            self.num_tries += 1
            if self.conf['go_back'] == 'True':
                print('WORKER %d: loading previous model' % self.task_index)
                self.validation_hook.restore()
            else:
                self.validated_step = self.global_step
            if self.conf['valid_adapt'] == 'True':
                print('WORKER %d: halving learning rate' % self.task_index)
                self.learning_rate_fact /= 2
                self.validation_hook.save()
        else:
            if self.conf['reset_tries'] == 'True':
                self.num_tries = 0
            self.validated_step = self.global_step
            self.best_validation = validation_loss
            self.validation_hook.save()
        return False

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
        self.validation_history = []
        num_steps = outputs['num_steps']
        # MonitoredTrainingSession(checkpoint_dir=expdir/logdir) restores the latest checkpoint
        # when one exists (reference trainer.py:625-633) and saves one every 600 s
        ckpt = None if self.expdir is None else os.path.join(self.expdir, 'logdir', 'model.ckpt')
        if ckpt is not None and os.path.exists(ckpt):
            self._ensure_variables()
            self.load_state(torch.load(ckpt, weights_only=False))
            print('WORKER %d: resumed from %s at step %d' % (self.task_index, ckpt, self.global_step))
        last_ckpt = time.time()
        while self.global_step < num_steps:
            if (self.evaluator is not None
                    and self.global_step - self.validated_step >= int(self.conf['valid_frequency'])):
                self._ensure_variables()
                if self._validation_point():
                    break
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
            on_gpu = torch.cuda.is_available()
            mem_used = torch.cuda.max_memory_allocated() if on_gpu else 0
            mem_total = torch.cuda.get_device_properties(0).total_memory if on_gpu else 0
            print(('WORKER %d: step %d/%d loss: %f, learning rate: %f \n\t time elapsed: %f sec'
                   '\n\t peak memory usage: %d/%d MB')
                  % (self.task_index, self.global_step, num_steps, loss_value, self.last_lr,
                     time.time() - start, mem_used / 1e6, mem_total / 1e6))
            history.append((self.global_step, loss_value, self.last_lr))
            self.global_step += 1
            every = getattr(self, 'checkpoint_steps', None)
            if is_chief and ckpt is not None and (
                    time.time() - last_ckpt > CHECKPOINT_SECS or (every and self.global_step % every == 0)):
                self.save_checkpoint(ckpt)
                last_ckpt = time.time()
        if is_chief and self.expdir is not None:
            self.save_checkpoint(ckpt)
            self.save()
        return history

    def save_checkpoint(self, path):
        os.makedirs(os.path.dirname(path), exist_ok=True)
        torch.save(self.state(), path + '.tmp')
        os.replace(path + '.tmp', path)

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
