"""The input pipeline over the reference's on-disk format
(nabu/processing/input_pipeline.py:9-202, consumed by Trainer._data, trainers/trainer.py:285-423 and
Evaluator.evaluate, evaluators/evaluator.py:62-123).

The reference builds TF queue runners: a shuffled filename queue, one reader per data set,
`bucket_by_sequence_length` (or `tf.train.batch`) with dynamic padding.  Here the same stream of
batches is produced by plain Python + numpy as a random-access batch source with the interface
of processing.synthetic.SyntheticData (`num_batches()`, `batch(step)`, `validation(...)`), i.e.
the A0 batch contract of SURVEY.md 8(a):

* utterances present in every data set are kept (`get_filenames`);
* each epoch visits them in a freshly shuffled order (seeded: default_rng([seed, epoch]));
* with numbuckets > 1 an utterance goes to the bucket of its FIRST data set's sequence length
  (boundaries from the greedy `bucket_boundaries` on the length histogram); a bucket emits a batch
  as soon as it holds its batch size (variable_batch_size: max(int(batch_size*b0/b), 1)); leftovers
  stay queued into the next epoch like in the reference's never-ending queues;
* batches are zero padded to the longest member (dynamic_pad);
* the next batch is assembled by a background thread while the current one trains (the reference's
  queue runners), and bucketing takes the utterance lengths from the records' length prefixes instead
  of decoding every file before the first step."""
import os
import threading
from concurrent.futures import ThreadPoolExecutor

import numpy as np

from nabu_amd.processing.tfreaders import tfreader_factory


def get_filenames(dataconfs):
    '''the utterances found in all data sets (reference input_pipeline.py:9-55)

    Args:
        dataconfs: the database configurations as a list (one per data set) of lists of sections
    Returns:
        - a list of tuples with the filenames of an utterance, one per data set
        - a list containing the names'''
    files = []
    for dataconfset in dataconfs:
        setfiles = dict()
        for i, dataconf in enumerate(dataconfset):
            with open(os.path.join(dataconf['dir'], 'pointers.scp')) as fid:
                for line in fid:
                    (n, f) = line.strip().split('\t')
                    setfiles['%s-%d' % (n, i)] = f
        files.append(setfiles)
    elements, names = [], []
    for name in files[0]:
        if all(name in setfile for setfile in files):
            elements.append(tuple(setfile[name] for setfile in files))
            names.append(name)
        else:
            print('%s was not found in all sets of data, ignoring this example' % name)
    return elements, names


def bucket_boundaries(histogram, numbuckets):
    '''bucket boundaries that divide the number of elements uniformly — the reference's greedy
    algorithm (input_pipeline.py:176-202)'''
    boundaries = [0] * numbuckets
    for i in range(numbuckets - 1):
        numelements = int(histogram[boundaries[i]:].sum() / (numbuckets - i))
        if numelements == 0:
            print('%d buckets could not be reached, using %d buckets' % (numbuckets, i))
        j = boundaries[i] + 1
        while (j + 1 < len(histogram) and
               abs(histogram[boundaries[i]:j].sum() - numelements) >=
               abs(histogram[boundaries[i]:j + 1].sum() - numelements)):
            j += 1
        boundaries[i + 1] = j
    return boundaries[1:]


class RecordData(object):
    '''batch source over TFRecord data sets'''

    def __init__(self, input_names, input_dataconfs, target_names, target_dataconfs, batch_size,
                 numbuckets=1, variable_batch_size=False, shuffle=True, seed=0):
        self.input_names, self.target_names = list(input_names), list(target_names)
        dataconfs = list(input_dataconfs) + list(target_dataconfs)
        self.elements, self.names = get_filenames(dataconfs)
        self.readers = []
        for dataconfset in dataconfs:
            types = [d['type'] for d in dataconfset]
            if len(set(types)) > 1:
                raise Exception('all data types in a set must be the same')
            self.readers.append(tfreader_factory.factory(types[0])([d['dir'] for d in dataconfset]))
        histogram = self.readers[0].metadata['sequence_length_histogram']
        self.max_length = histogram.size
        self.batch_size = int(batch_size)
        self.shuffle, self.seed = shuffle, int(seed)
        if numbuckets > 1:                                   # input_pipeline.py:121-158
            self.boundaries = bucket_boundaries(histogram, numbuckets)
            if variable_batch_size:
                self.batch_sizes = [max(int(self.batch_size * self.boundaries[0] / b), 1)
                                    for b in self.boundaries + [histogram.size]]
                numutt = ([histogram[:self.boundaries[0]].sum()] +
                          [histogram[self.boundaries[i]:b].sum() for i, b in enumerate(self.boundaries[1:])] +
                          [histogram[self.boundaries[-1]:].sum()])
                self.num_steps = int((np.array(numutt) / np.array(self.batch_sizes)).sum())
            else:
                self.batch_sizes = [self.batch_size] * (len(self.boundaries) + 1)
                self.num_steps = int(histogram.sum() / self.batch_size)
        else:
            self.boundaries = []
            self.batch_sizes = [self.batch_size]
            self.num_steps = int(histogram.sum() / self.batch_size)
        self._lengths = None
        self._pool = ThreadPoolExecutor(max_workers=1)
        self._lock = threading.Lock()
        self._ahead = {}           # step -> Future of its batch
        self._last_step = None
        self._epochs = []          # per epoch: list of batches (lists of utterance indices)
        self._carry = [[] for _ in self.batch_sizes]
        self._cache = {}

    # -- the reference reports this as the number of steps of an epoch
    def num_batches(self):
        return self.num_steps

    def _first_lengths(self):
        if self._lengths is None:
            self._lengths = np.array([self.readers[0].sequence_length(e[0]) for e in self.elements], np.int64)
        return self._lengths

    def _schedule_epoch(self):
        '''simulate the bucketing queues over one pass through the (shuffled) utterances'''
        epoch = len(self._epochs)
        order = np.arange(len(self.elements))
        if self.shuffle:
            order = np.random.default_rng([self.seed, epoch]).permutation(len(self.elements))
        batches = []
        lengths = self._first_lengths() if self.boundaries else None
        for u in order:
            b = int(np.searchsorted(self.boundaries, lengths[u], side='right')) if self.boundaries else 0
            self._carry[b].append(int(u))
            if len(self._carry[b]) == self.batch_sizes[b]:
                batches.append(self._carry[b])
                self._carry[b] = []
        self._epochs.append(batches)

    def _indices(self, step):
        total = 0
        e = 0
        while True:
            if e == len(self._epochs):
                self._schedule_epoch()
                if not self._epochs[e] and not any(self._epochs[max(0, e - 3):]):
                    raise Exception('the data set is too small to fill a single batch')
            if step < total + len(self._epochs[e]):
                return self._epochs[e][step - total]
            total += len(self._epochs[e])
            e += 1

    def _read(self, u):
        if u not in self._cache:
            if len(self._cache) > 4096:
                self._cache.clear()
            self._cache[u] = [r(f) for r, f in zip(self.readers, self.elements[u])]
        return self._cache[u]

    def batch(self, step):
        '''batch number `step` of the never-ending stream (A0 contract, numpy).  The batch the caller
        will most likely ask for next (same stride as the last two requests) is read and padded by a
        background thread meanwhile.'''
        fut = self._ahead.pop(step, None)
        out = fut.result() if fut is not None else self._assemble(step)
        stride = step - self._last_step if self._last_step is not None and step > self._last_step else 1
        self._last_step = step
        nxt = step + stride
        self._ahead = {nxt: self._ahead[nxt]} if nxt in self._ahead else {}      # drop guesses that were wrong
        if nxt not in self._ahead:
            self._ahead[nxt] = self._pool.submit(self._assemble, nxt)
        return out

    def _assemble(self, step):
        with self._lock:                                   # schedule, carries and cache are shared state
            return self._assemble_locked(step)

    def _assemble_locked(self, step):
        utts = [self._read(u) for u in self._indices(step)]
        names = self.input_names + self.target_names
        out = dict(inputs={}, input_seq_length={}, targets={}, target_seq_length={})
        for i, name in enumerate(names):
            arrays = [u[i][0] for u in utts]
            lens = np.array([u[i][1] for u in utts], np.int32)
            pad = np.zeros((len(arrays), int(lens.max())) + arrays[0].shape[1:], arrays[0].dtype)
            for j, a in enumerate(arrays):
                pad[j, :a.shape[0]] = a
            key, lkey = ('inputs', 'input_seq_length') if i < len(self.input_names) else ('targets', 'target_seq_length')
            out[key][name] = pad
            out[lkey][name] = lens
        return out


def from_sections(dataconf, input_names, input_sections, target_names, target_sections, **kwargs):
    '''build a RecordData from database.conf section names (trainer.py:289-318, evaluator.py:37-60)'''
    def confs(sectionsets):
        return [[dict(dataconf.items(section)) for section in sectionset] for sectionset in sectionsets]
    return RecordData(input_names, confs(input_sections), target_names, confs(target_sections), **kwargs)
