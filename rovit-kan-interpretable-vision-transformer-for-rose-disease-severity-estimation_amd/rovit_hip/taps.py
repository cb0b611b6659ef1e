"""Explainability taps (not on the hot path)."""


def attention_outputs(model, x):
    raise NotImplementedError('attention taps from the fused backbone are a later row of the scope table (f-4)')
