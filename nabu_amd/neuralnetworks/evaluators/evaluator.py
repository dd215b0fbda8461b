"""General evaluator (reference: nabu/neuralnetworks/evaluators/evaluator.py:9-146).

An evaluator measures a model on a validation set.  The reference builds a second input
pipeline from the database sections its conf names and returns (loss variable, update op,
number of batches); here ``evaluate()`` returns the same triple as plain Python objects:
a 1-element list holding the running loss, a function that folds validation batch ``i`` into
it, and the number of batches.  The validation data come from the dev sections of database.conf
named in the evaluator conf (processing.input_pipeline), or — for a synthetic batch source — from
``dataconf.validation(numbatches, batch_size)``."""
from abc import ABCMeta, abstractmethod

from nabu_amd.tools.default_conf import apply_defaults, defaults_path


class Evaluator(object, metaclass=ABCMeta):
    '''the general evaluator class'''

    def __init__(self, conf, dataconf, model):
        '''Args:
            conf: the evaluator configuration as a ConfigParser ([evaluator] section)
            dataconf: the batch source the trainer uses (must offer ``validation``)
            model: the model to be evaluated'''
        self.conf = dict(conf.items('evaluator'))
        apply_defaults(self.conf, defaults_path(__file__, self))
        self.model = model
        if hasattr(dataconf, 'validation'):
            # synthetic batch source: a disjoint set with the same statistics
            self.data = dataconf.validation(int(self.conf['numbatches']), int(self.conf['batch_size']))
            return
        if not hasattr(dataconf, 'has_section'):
            raise Exception('the evaluator needs a batch source with validation() or a database conf')
        # the reference's on-disk data (reference evaluator.py:37-90): the evaluator conf links the
        # model's input names and the target names to (dev) sections of database.conf; utterances in
        # file order, one bucket, len(set) // batch_size batches
        from nabu_amd.processing import input_pipeline
        input_names = list(self.model.input_names)
        target_names = [n for n in self.conf['targets'].split(' ') if n]
        self.data = input_pipeline.from_sections(
            dataconf, input_names, [self.conf[i].split(' ') for i in input_names],
            target_names, [self.conf[o].split(' ') for o in target_names],
            batch_size=int(self.conf['batch_size']), numbuckets=1, shuffle=False)
        nb = len(self.data.elements) // int(self.conf['batch_size'])
        self.data.num_steps = nb
        self.data.bounded = True          # no look-ahead batch past the last validation batch

    def evaluate(self):
        '''Returns:
            - the loss as a 1-element list (the reference's loss variable)
            - update(i): folds validation batch i into the loss (the reference's update op)
            - the number of batches in the validation set'''
        loss = [0.0]
        self.reset()

        def update(i):
            self.update_loss(loss, self.data.batch(i))
        return loss, update, self.data.num_batches()

    @abstractmethod
    def reset(self):
        '''re-initialise the accumulators (the reference's init_validation op)'''

    @abstractmethod
    def update_loss(self, loss, batch):
        '''fold one batch (A0 contract, numpy) into loss[0]'''
