"""Grammars for the Viterbi decoder -- host-side mirror of reference src/core/viterbi/grammar.py.

Only SingleTranscriptGrammar (reference grammar.py:196-217) is decoded with anywhere in the
reference (src/mucon/evaluators.py:148-150); NGram / PathGrammar / ModifiedPathGrammar have no
decode call site and are out of scope."""
import numpy as np


class Grammar(object):
    """Interface of reference grammar.py:6-35."""

    def score(self, context, label):
        return 0.0

    def n_classes(self):
        return 0

    def start_symbol(self):
        return -1

    def end_symbol(self):
        return -2

    def possible_successors(self, context):
        return set()

    def update_context(self, context, label):
        return context + (label,)


class SingleTranscriptGrammar(Grammar):
    """Generates exactly one transcript: after (start, a_0 .. a_{n-1}) only a_n may follow, then the
    end symbol (reference grammar.py:196-217).  The HIP decoder consumes `.transcript` directly."""

    def __init__(self, transcript, n_classes):
        self.num_classes = n_classes
        self.transcript = [int(x) for x in transcript]

    def n_classes(self):
        return self.num_classes

    def possible_successors(self, context):
        n = len(context) - 1
        if n < 0 or context[0] != self.start_symbol() or list(context[1:]) != self.transcript[:n]:
            return set()
        return {self.transcript[n]} if n < len(self.transcript) else ({self.end_symbol()} if n == len(self.transcript) else set())

    def score(self, context, label):
        return 0.0 if label in self.possible_successors(context) else -np.inf
