"""EDEncoder factory (reference: models/ed_encoders/ed_encoder_factory.py:4-29)."""


def factory(encoder):
    '''get an EDEncoder class by its recipe name'''
    if encoder == 'listener':
        from nabu_amd.neuralnetworks.models.ed_encoders import listener
        return listener.Listener
    elif encoder == 'dblstm':
        from nabu_amd.neuralnetworks.models.ed_encoders import dblstm
        return dblstm.DBLSTM
    elif encoder in ('dummy_encoder', 'dnn', 'hotstart_encoder'):
        raise Exception('encoder type %s is outside the MI355X hot path (SURVEY.md 2.1 row 2)'
                        % encoder)
    else:
        raise Exception('undefined encoder type: %s' % encoder)
