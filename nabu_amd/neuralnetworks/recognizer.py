"""Recognizer (reference: nabu/neuralnetworks/recognizer.py:12-150): decodes a data set with a
trained model and stores the results under <expdir>/decoded.

The reference feeds a TF queue of file names through input_pipeline with
allow_smaller_final_batch; here the same sections of database.conf are read through
processing.input_pipeline.RecordData in file order, ceil(N / batch_size) batches, the last one
smaller.  Decoding itself is the decoder's business (device kernels, decoders/)."""
import math
import os
import shutil

import numpy as np
import torch

from nabu_amd.autodiff import SeqLen
from nabu_amd.neuralnetworks.decoders import decoder_factory
from nabu_amd.tools.default_conf import apply_defaults, defaults_path


class Recognizer(object):
    '''a Recognizer uses a model to decode a data set; results go to disk'''

    def __init__(self, model, conf, dataconf, expdir):
        '''Args:
            model: the model to decode with
            conf: the recognizer configuration (ConfigParser: [recognizer] and [decoder])
            dataconf: the database configuration (ConfigParser)
            expdir: the experiments directory'''
        from nabu_amd.processing import input_pipeline
        self.conf = dict(conf.items('recognizer'))
        apply_defaults(self.conf, defaults_path(__file__, self))
        self.expdir = expdir
        self.model = model
        self.batch_size = int(self.conf['batch_size'])
        self.decoder = decoder_factory.factory(conf.get('decoder', 'decoder'))(conf, model)
        input_names = list(model.input_names)
        self.data = input_pipeline.from_sections(
            dataconf, input_names, [self.conf[i].split(' ') for i in input_names], [], [],
            batch_size=self.batch_size, numbuckets=1, shuffle=False)
        self.names = list(self.data.names)
        self.numbatches = int(math.ceil(float(len(self.data.elements)) / self.batch_size))

    def _batch(self, i):
        '''utterances [i*batch_size, (i+1)*batch_size) in file order (the last batch may be smaller)'''
        idx = range(i * self.batch_size, min((i + 1) * self.batch_size, len(self.data.elements)))
        utts = [self.data._read(u) for u in idx]
        inputs, lengths = {}, {}
        for k, name in enumerate(self.data.input_names):
            arrays = [u[k][0] for u in utts]
            lens = np.array([u[k][1] for u in utts], np.int32)
            pad = np.zeros((len(arrays), int(lens.max())) + arrays[0].shape[1:], np.float32)
            for j, a in enumerate(arrays):
                pad[j, :a.shape[0]] = a
            inputs[name], lengths[name] = pad, lens
        return inputs, lengths

    def recognize(self):
        '''load <expdir>/model/network.ckpt.npz if present, decode everything, write to <expdir>/decoded'''
        ckpt = os.path.join(self.expdir, 'model', 'network.ckpt.npz')
        if os.path.exists(ckpt):
            with np.load(ckpt) as state:
                self.model.store.restore_from({k: state[k] for k in state.files})
        directory = os.path.join(self.expdir, 'decoded')
        if os.path.isdir(directory):
            shutil.rmtree(directory)
        os.makedirs(directory)
        dev = torch.device('cuda', torch.cuda.current_device())
        for i in range(self.numbatches):
            inputs, lengths = self._batch(i)
            outputs = self.decoder({n: torch.as_tensor(a).to(dev) for n, a in inputs.items()},
                                   {n: SeqLen.wrap(a, dev) for n, a in lengths.items()})
            names = self.names[i * self.batch_size:(i + 1) * self.batch_size]
            names = ['-'.join(name.split('-')[:-1]) for name in names]     # drop the data-set index
            self.decoder.write(outputs, directory, names)
        return directory
