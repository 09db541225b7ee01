from .grammar import Grammar, SingleTranscriptGrammar  # noqa: F401
from .length_model import LengthModel, PoissonModel  # noqa: F401
from .viterbi import Viterbi  # noqa: F401
