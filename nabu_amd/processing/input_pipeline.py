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


def _pointer_table(sections):
    """utterance key -> record file of one data set; a data set may be spread over several
    sections (directories), the key carries the section index so equal names do not collide"""
    table = {}
    for index, section in enumerate(sections):
        with open(os.path.join(section['dir'], 'pointers.scp')) as scp:
            for row in scp:
                utt, path = row.strip().split('\t')
                table['%s-%d' % (utt, index)] = path
    return table


def get_filenames(dataconfs):
    """The utterances that every data set holds (reference input_pipeline.py:9-55).

    dataconfs: one list of database sections per data set.  Returns (per-utterance tuples of record
    files, one entry per data set; the utterance keys), in the order of the first data set."""
    tables = [_pointer_table(sections) for sections in dataconfs]
    shared = set(tables[0]).intersection(*tables[1:])
    for key in tables[0]:
        if key not in shared:
            print('%s was not found in all sets of data, ignoring this example' % key)
    names = [key for key in tables[0] if key in shared]
    return [tuple(t[key] for t in tables) for key in names], names


def bucket_boundaries(histogram, numbuckets):
    """Length boundaries that spread the utterances evenly over numbuckets buckets — the greedy rule
    of the reference (input_pipeline.py:176-202): every bucket aims at an equal share of what is
    left and its end moves right for as long as the next length brings the count at least as close."""
    counts = np.concatenate([[0], np.cumsum(np.asarray(histogram))])     # counts[j] = elements shorter than j
    top = len(histogram) - 1
    edges, start = [], 0
    for done in range(numbuckets - 1):
        below = counts[min(start, top + 1)]
        share = int((counts[-1] - below) / (numbuckets - done))
        if share == 0:
            print('%d buckets could not be reached, using %d buckets' % (numbuckets, done))
        miss = np.abs(counts[start + 1:top + 1] - below - share)          # miss[k]: bucket ends at start + 1 + k
        closer = np.nonzero(miss[:-1] < miss[1:])[0]                      # first end the next length does not improve
        end = start + 1 + (int(closer[0]) if len(closer) else max(len(miss) - 1, 0))
        edges.append(end)
        start = end
    return edges


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
        self.lookahead = True
        self.bounded = False       # bounded consumers (validation) set it: no look-ahead past the last batch
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
        hit = self._cache.get(u)
        if hit is None:
            hit = [r(f) for r, f in zip(self.readers, self.elements[u])]
            if len(self._cache) > 4096:
                self._cache.clear()
            self._cache[u] = hit
        return hit

    def batch(self, step):
        '''batch number `step` of the never-ending stream (A0 contract, numpy).  The batch the caller
        will most likely ask for next (same stride as the last two requests) is read and padded by a
        background thread meanwhile.'''
        fut = self._ahead.pop(step, None)
        for stale in self._ahead.values():                 # guesses that were wrong: do not let them run
            stale.cancel()
        self._ahead = {}
        out = fut.result() if fut is not None else self._assemble(step)
        stride = step - self._last_step if self._last_step is not None and step > self._last_step else 1
        self._last_step = step
        nxt = step + stride
        if self.lookahead and self._pool is not None and not (self.bounded and nxt >= self.num_steps):
            self._ahead[nxt] = self._pool.submit(self._assemble, nxt)
        return out

    def close(self):
        """stop the look-ahead thread (pending guesses are dropped)"""
        pool, self._pool = self._pool, None
        for fut in self._ahead.values():
            fut.cancel()
        self._ahead = {}
        if pool is not None:
            pool.shutdown(wait=False)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _assemble(self, step):
        with self._lock:                                   # the schedule and the carries are shared state
            indices = list(self._indices(step))
        # file reads and padding run outside the lock: a wrong look-ahead guess does not hold up the
        # batch that was actually asked for (the utterance cache takes whole entries, a lost race
        # only reads a file twice)
        return self._pad(indices)

    def _pad(self, indices):
        utts = [self._read(u) for u in indices]
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
