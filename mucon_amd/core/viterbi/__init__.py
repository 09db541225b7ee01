from .grammar import Grammar, SingleTranscriptGrammar  # noqa: F401
from .length_model import LengthModel, PoissonModel, PoissonRows, poisson_rows_for_many  # noqa: F401
from .viterbi import NoHypothesisError, ShortSequenceError, Viterbi  # noqa: F401
