"""Base of the per-utterance TFRecord writers (on-disk layout of nabu/processing/tfwriters/tfwriter.py:9-55):
utterance number N goes to <datadir>/data/fileN, and <datadir>/pointers.scp gains one line
``name<TAB>path`` so that readers and `get_filenames` can find it."""
import os
from abc import ABCMeta, abstractmethod

from nabu_amd.processing import tfrecord


class TfWriter(object, metaclass=ABCMeta):
    """Subclasses turn one utterance into a serialized tf.train.Example (`_get_example`)."""

    def __init__(self, datadir):
        self.write_dir = os.path.join(datadir, 'data')
        os.makedirs(self.write_dir)                     # also creates datadir; refuses to overwrite a set
        self.scp_file = os.path.join(datadir, 'pointers.scp')
        self.filenum = 0

    def write(self, data, name):
        path = os.path.join(self.write_dir, 'file%d' % self.filenum)
        tfrecord.write_records(path, [self._get_example(data)])
        with open(self.scp_file, 'a') as scp:
            scp.write(name + '\t' + path + '\n')
        self.filenum += 1

    @abstractmethod
    def _get_example(self, data):
        """bytes of the tf.train.Example holding `data`"""
