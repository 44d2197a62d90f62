from .classifier import ClassifierModel
from .deep_classifiers import CNNClassifier, CNNRNNClassifier
from .simple_classifiers import LogisticRegressionClassifier, ShallowNNClassifier
from .synthesis_models import SynthesisModel, SynthesisModelCNN, SynthesisLite
from .synthesis_trainer import SynthesisTrainer, compute_mcd
from .classifier_factory import get_classifier_by_name
from .classifier_trainer import ClassifierTrainer
