"""Campaign drivers over ``classify_*_with_faults`` -- same classes, methods, arguments and result
files as the reference's ``bnn/faults/faults.py``, written for this package.

``FaultTest.run_test`` builds a fresh classifier for every run (``load_parameters`` again: the
faults of the previous run are gone), injects ``num_flips`` upsets of ``word_size`` adjacent bits
while the input set is classified, and reports the accuracy of each run.  ``NetworkTest`` sweeps
fault counts x {any, weight, threshold} x {bit, word}, writes the raw accuracies and the statistics
(min / max / average, "effective" runs = runs whose accuracy differs from the control) as JSON.

On the MI355X runtime a run of 10 000 CIFAR-10 images with 100 upsets takes about 5 ms plus the
3 ms parameter reload (tools/fault_campaign_rate.py).
"""
from .. import bnn as _bnn
from .. import util


class FaultTest:
    class TargetType:
        @staticmethod
        def any():
            return -1

        @staticmethod
        def weights():
            return 0

        @staticmethod
        def thresholds():
            return 1

    def __init__(self, classifier_cls, network, dataset, input_file, labels, runtime=_bnn.RUNTIME_HW):
        self.classifier_cls = classifier_cls
        self.network = network
        self.dataset = dataset
        self.input_file = input_file  # a file path, or a list of PIL images for the non-CIFAR picture sets
        self.labels = labels
        self.runtime = runtime

    def run_test(self, num_runs, num_flips, word_size=1, target_type=-1, target_layers=()):
        """-> (results per run, usec per image per run, accuracy per run)"""
        target_layers = list(target_layers)
        results, times, accuracies = [], [], []
        for i in range(num_runs):
            classifier = self.classifier_cls(self.network, self.dataset, self.runtime)
            print("{}-{} run {} of {} (flipping {}{} {}(s) in {})".format(
                self.network, self.dataset, i + 1, num_runs, num_flips,
                " weight" if target_type == 0 else " threshold" if target_type == 1 else "",
                "word" if word_size > 1 else "bit",
                "any layer" if not target_layers else "layer(s) {}".format(target_layers)))
            if self.dataset == "cifar10":
                got = classifier.classify_cifars_with_faults(self.input_file, num_flips, word_size, target_type, target_layers)
            elif self.dataset == "mnist":
                got = classifier.classify_mnists_with_faults(self.input_file, num_flips, word_size, target_type, target_layers)
            else:
                got = classifier.classify_images_with_faults(self.input_file, num_flips, word_size, target_type, target_layers)
            results.append(got.tolist())
            times.append(classifier.usecPerImage)
            accuracies.append(util.calculate_accuracy(results[-1], self.labels))
            print("Accuracy:", accuracies[-1])
            print()
        return (results, times, accuracies)


class CNVFaultTest(FaultTest):
    def __init__(self, network, dataset, input_file, labels, runtime=_bnn.RUNTIME_HW):
        super().__init__(_bnn.CnvClassifier, network, dataset, input_file, labels, runtime)

    @classmethod
    def CIFARTest(cls, network, input_file, labels):
        return cls(network, "cifar10", input_file, labels)

    @classmethod
    def SVHNTest(cls, network, input_file, labels):
        return cls(network, "streetview", input_file, labels)

    @classmethod
    def GTSRBTest(cls, network, input_file, labels):
        return cls(network, "road-signs", input_file, labels)


class LFCFaultTest(FaultTest):
    def __init__(self, network, dataset, input_file, labels, runtime=_bnn.RUNTIME_HW):
        super().__init__(_bnn.LfcClassifier, network, dataset, input_file, labels, runtime)

    @classmethod
    def MNISTTest(cls, network, input_file, labels):
        return cls(network, "mnist", input_file, labels)


class NetworkTest:
    class TestType:
        def __init__(self, target_type, word_size):
            self.target_type = target_type
            self.word_size = word_size
            where = {-1: "any", 0: "weight"}.get(target_type, "threshold")
            self.name = where + " " + ("word" if word_size > 1 else "bit")

        @classmethod
        def any_bit(cls):
            return cls(FaultTest.TargetType.any(), 1)

        @classmethod
        def any_word(cls, word_size=8):
            return cls(FaultTest.TargetType.any(), word_size)

        @classmethod
        def weight_bit(cls):
            return cls(FaultTest.TargetType.weights(), 1)

        @classmethod
        def weight_word(cls, word_size=8):
            return cls(FaultTest.TargetType.weights(), word_size)

        @classmethod
        def threshold_bit(cls):
            return cls(FaultTest.TargetType.thresholds(), 1)

        @classmethod
        def threshold_word(cls, word_size=8):
            return cls(FaultTest.TargetType.thresholds(), word_size)

    def __init__(self, fault_test):
        self.fault_test = fault_test
        self.control = None

    def _run_control(self):
        print("Running", self.fault_test.network + "-" + self.fault_test.dataset, "control test")
        _, _, accuracy = self.fault_test.run_test(num_runs=1, num_flips=0)
        self.control = accuracy[0]

    def _raw(self, name, num_runs, num_flips, layers, accuracies):
        return {"network": self.fault_test.network, "dataset": self.fault_test.dataset, "run count": num_runs,
                "flips": num_flips, "control": self.control, "layers": list(layers), "results": {name: accuracies}}

    def _stats(self, merged):
        out = dict(merged)
        out["results"] = {}
        for name, runs in merged["results"].items():
            effective = [a for a in runs if a != merged["control"]]
            entry = {"runs": {"all": runs, "effective": effective}, "effective count": len(effective),
                     "min accuracy": min(runs), "max accuracy": max(runs)}
            if effective:
                entry["avg accuracy"] = sum(runs) / len(runs)
                entry["avg effective accuracy"] = sum(effective) / len(effective)
                entry["accuracy delta"] = merged["control"] - entry["avg accuracy"]
                entry["effective accuracy delta"] = merged["control"] - entry["avg effective accuracy"]
            else:
                entry["avg accuracy"] = merged["control"]
            out["results"][name] = entry
        return out

    def _run_tests(self, folder, num_runs, num_flips, test_types, target_layers):
        raw = []
        for test in test_types:
            _, _, accuracies = self.fault_test.run_test(num_runs, num_flips, test.word_size, test.target_type, target_layers)
            raw.append(self._raw(test.name, num_runs, num_flips, target_layers, accuracies))
            util.write_dict_to_file("{}/temp/{}_results_{}.json".format(folder, self.fault_test.network, test.name.replace(" ", "-")), raw[-1])
        return self._stats(util.dict_of_dicts_merge(*raw))

    def test_network(self, output_folder, num_runs, flip_counts, test_types, target_layers=()):
        """one statistics file per fault count under output_folder/<network>/<dataset>/<n>flips/"""
        output_folder = "{}/{}/{}/".format(output_folder, self.fault_test.network, self.fault_test.dataset)
        if self.control is None:
            self._run_control()
        for num_flips in flip_counts:
            folder = "{}/{}flips/".format(output_folder, num_flips)
            stats = self._run_tests(folder, num_runs, num_flips, test_types, target_layers)
            name = "{}/{}_{}".format(folder, self.fault_test.network, self.fault_test.dataset)
            name += "_stats_layer{}.json".format(list(target_layers)) if len(target_layers) > 0 else "_stats.json"
            util.write_dict_to_file(name, stats)

    def comprehensive_test(self, output_folder, num_runs, flip_counts, target_layers=()):
        """all six combinations of {any, weight, threshold} x {bit, 8-bit word}.  (The reference's version
        forgets to pass an output folder on to test_network, faults.py:254-261; here it is the first argument.)"""
        T = NetworkTest.TestType
        self.test_network(output_folder, num_runs, flip_counts,
                          [T.any_bit(), T.any_word(), T.weight_bit(), T.weight_word(), T.threshold_bit(), T.threshold_word()],
                          target_layers)
