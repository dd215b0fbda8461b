"""Evaluator factory (reference: nabu/neuralnetworks/evaluators/evaluator_factory.py:4-24)."""


def factory(evaluator):
    '''gets an evaluator class

    Args:
        evaluator: the evaluator type
    Returns:
        an evaluator class'''
    if evaluator == 'loss_evaluator':
        from nabu_amd.neuralnetworks.evaluators import loss_evaluator
        return loss_evaluator.LossEvaluator
    elif evaluator == 'decoder_evaluator':
        raise Exception('decoder_evaluator (beam-search decoding + error rate) is inference '
                        '(SURVEY.md 8(f) row 4), not part of the training hot path')
    else:
        raise Exception('Undefined evaluator type: %s' % evaluator)
