"""Synthetic batches with the exact batch contract the reference's input pipeline
hands to the model (Trainer._data, reference trainers/trainer.py:285-423; readers
audio_feature_reader.py:64-78, string_reader.py:76-105, string_reader_eos.py:82-111):

  inputs['features']            [B,T_max,D] float32, zero padded past the length
  input_seq_length['features']  [B] int32
  targets['text']               [B,L_max] int32 label ids in 0..C-2, zero padded
                                (string_eos targets carry EOS = C-1 at position len-1)
  target_seq_length['text']     [B] int32

Generator: numpy PCG64, independent streams per tensor
default_rng([seed, step, k]) with k = 0 lengths, 1 features, 2 targets, so a
batch is a pure function of (seed, step)."""
import numpy as np


class SyntheticData(object):
    def __init__(self, batch_size, max_frames, feature_dim, num_labels=39, min_frames=None,
                 min_labels=10, max_labels=40, eos=False, time_reduction=1, seed=1234,
                 batches_per_epoch=100, input_name='features', target_name='text'):
        self.B, self.T, self.D = int(batch_size), int(max_frames), int(feature_dim)
        self.num_labels = int(num_labels)
        self.min_frames = self.T if min_frames is None else int(min_frames)
        self.min_labels, self.max_labels = int(min_labels), int(max_labels)
        self.eos = bool(eos)
        self.time_reduction = int(time_reduction)      # encoder frames = ceil(len / this)
        self.seed = int(seed)
        self.batches_per_epoch = int(batches_per_epoch)
        self.input_name, self.target_name = input_name, target_name

    def num_batches(self):
        return self.batches_per_epoch

    def validation(self, numbatches, batch_size=None):
        '''a disjoint batch source with the same statistics: the counterpart of the dev
        sections of the reference's database.conf (evaluators/evaluator.py:37-60)'''
        v = SyntheticData(self.B if batch_size is None else batch_size, self.T, self.D, self.num_labels,
                          self.min_frames, self.min_labels, self.max_labels, self.eos, self.time_reduction,
                          self.seed + 1000003, numbatches, self.input_name, self.target_name)
        return v

    def batch(self, step):
        B, T, D = self.B, self.T, self.D
        r_len = np.random.default_rng([self.seed, step, 0])
        r_feat = np.random.default_rng([self.seed, step, 1])
        r_tgt = np.random.default_rng([self.seed, step, 2])
        lens = r_len.integers(self.min_frames, T + 1, B).astype(np.int32)
        lens[0] = T                                       # at least one full-length utterance
        feats = r_feat.standard_normal((B, T, D), dtype=np.float32)
        feats *= (np.arange(T)[None, :, None] < lens[:, None, None])
        Lcap = self.max_labels + (1 if self.eos else 0)
        targets = np.zeros((B, Lcap), np.int32)
        tlen = np.zeros(B, np.int32)
        for b in range(B):
            enc = -(-int(lens[b]) // self.time_reduction)
            while True:
                L = int(r_tgt.integers(self.min_labels, self.max_labels + 1))
                lab = r_tgt.integers(0, self.num_labels, L)
                rep = int(np.sum(lab[1:] == lab[:-1]))
                if self.eos or L + rep <= enc:            # CTC feasibility
                    break
            targets[b, :L] = lab
            tlen[b] = L
            if self.eos:
                targets[b, L] = self.num_labels           # EOS = C-1 (string_reader_eos.py:60,101)
                tlen[b] = L + 1
        return dict(inputs={self.input_name: feats}, input_seq_length={self.input_name: lens},
                    targets={self.target_name: targets}, target_seq_length={self.target_name: tlen})
