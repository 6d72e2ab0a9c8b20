"""Same names, arguments and return values as the reference's ``bnn/util/util.py`` (the callers'
side of the classifiers: what its notebooks and ``bnn.faults`` use), written for this package.

Differences: ``dict_to_str`` pretty-prints with ``json`` (the reference pipes the JSON text
through ``yapf``, which is only layout); loaders raise ``FileNotFoundError`` with the missing path.
"""
import copy
import csv
import json
import os


def calculate_accuracy(results, labels):
    """percentage of ``labels`` that ``results`` reproduces (util.py:11-16)"""
    labels = list(labels)
    if not labels:
        return 0.0
    right = sum(1 for got, want in zip(results, labels) if got == want)
    return right * 100 / len(labels)


def dict_of_dicts_merge(*dicts):
    """union of dictionaries; values that are dictionaries on both sides are merged recursively,
    otherwise the first occurrence of a key wins (util.py:19-28)"""
    out = {}
    for item in dicts:
        for key, value in item.items():
            if key not in out:
                out[key] = copy.deepcopy(value)
            elif isinstance(out[key], dict) and isinstance(value, dict):
                out[key] = dict_of_dicts_merge(out[key], value)
    return out


def dict_to_str(d):
    return json.dumps(d, indent=4)


def write_dict_to_file(file_name, output_dict):
    folder = os.path.dirname(file_name)
    if folder:
        os.makedirs(folder, exist_ok=True)
    with open(file_name, "w+") as f:
        f.write(dict_to_str(output_dict))


def load_cifar10_testset(folder, num_images=10000):
    """(path of a CIFAR-10 binary file with the first ``num_images`` test records, their labels);
    a shortened copy ``test_batch_<n>.bin`` is written next to ``test_batch.bin`` when needed
    (util.py:44-69)"""
    num_images = min(num_images, 10000)
    full = os.path.join(folder, "test_batch.bin")
    path = full if num_images == 10000 else os.path.join(folder, "test_batch_{}.bin".format(num_images))
    if not os.path.exists(path):
        with open(full, "rb") as src, open(path, "wb") as dst:
            dst.write(src.read(3073 * num_images))
    with open(path, "rb") as f:
        data = f.read(3073 * num_images)
    labels = [data[i] for i in range(0, len(data) - 3072, 3073)]
    return (path, labels)


def load_gtsrb_testset(gt_file, images_folder, num_images=12630):
    """(list of opened PIL images, labels) from the GTSRB ground-truth CSV (``;`` separated,
    file name in column 0, class in column 7; util.py:73-93)"""
    from PIL import Image
    num_images = min(num_images, 12630)
    files, labels = [], []
    with open(gt_file) as f:
        rows = csv.reader(f, delimiter=";")
        next(rows)
        for row in rows:
            files.append(os.path.join(images_folder, row[0]))
            labels.append(int(row[7]))
    labels = labels[:num_images]
    images = []
    for name in files[:num_images]:
        img = Image.open(name)
        img.load()
        images.append(img)
    return (images, labels)


def load_svhn_testset(file, num_images=26032):
    """(list of PIL images, labels) from the SVHN ``test_32x32.mat``; labels are shifted by one to
    match the classifier's output -- class 1 for the digit 2 (util.py:97-118)"""
    import scipy.io as sio
    from PIL import Image
    num_images = min(num_images, 26032)
    data = sio.loadmat(file)
    labels = (data["y"] - 1).transpose().tolist()[0][:num_images]
    X = data["X"]
    images = [Image.fromarray(X[:, :, :, i]) for i in range(num_images)]
    return (images, labels)


def load_mnist_testset(folder, num_images=10000):
    """(path of ``t10k-images-idx3-ubyte``, the first ``num_images`` labels of
    ``t10k-labels-idx1-ubyte``) (util.py:122-134)"""
    num_images = min(num_images, 10000)
    idx3_path = os.path.join(folder, "t10k-images-idx3-ubyte")
    with open(os.path.join(folder, "t10k-labels-idx1-ubyte"), "rb") as f:
        f.read(8)  # magic, count (big endian)
        labels = list(f.read(num_images))
    return (idx3_path, labels)
