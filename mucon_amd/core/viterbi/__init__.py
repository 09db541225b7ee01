from .grammar import Grammar, SingleTranscriptGrammar  # noqa: F401
from .length_model import LengthModel, PoissonModel, PoissonParams, PoissonRows, poisson_params_for_many, poisson_rows_for_many  # noqa: F401
from .viterbi import NoHypothesisError, ShortSequenceError, Viterbi  # noqa: F401
