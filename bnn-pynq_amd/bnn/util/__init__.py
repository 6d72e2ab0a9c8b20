"""Helpers around the classifiers: accuracy, result dictionaries, test-set loaders
(mirror of the reference's ``bnn.util``, bnn/util/util.py)."""
from .util import (calculate_accuracy, dict_of_dicts_merge, dict_to_str, load_cifar10_testset,  # noqa: F401
                   load_gtsrb_testset, load_mnist_testset, load_svhn_testset, write_dict_to_file)

__version__ = "0.1"
