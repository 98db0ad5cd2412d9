"""Scalar summaries of the training driver: the counterparts of the reference's LossSummary and
PrecisionSummary (utils.py:151-199, 236-283) with the same tags, accumulation and per-epoch push --
written as JSON lines (`{"tag": ..., "value": ..., "step": ...}`) instead of TensorBoard event files
(there is no TensorFlow here; image summaries are cv2 drawing and stay out)."""
import json
import os


class SummaryWriter:
    """tf.summary.FileWriter stand-in: one scalars.jsonl under `logdir`."""

    def __init__(self, logdir):
        os.makedirs(logdir, exist_ok=True)
        self.path = os.path.join(logdir, 'scalars.jsonl')
        self._f = open(self.path, 'a')

    def add_scalar(self, tag, value, step):
        self._f.write(json.dumps({'tag': tag, 'value': float(value), 'step': int(step)}) + '\n')

    def flush(self):
        self._f.flush()

    def close(self):
        self._f.close()


class LossSummary:
    """utils.py:236-283: sample-weighted sums of the four losses over an epoch, pushed as
    `<sample_name>_<loss>_loss` = sum / num_samples."""
    loss_names = ['total', 'localization', 'confidence', 'l2']

    def __init__(self, writer, sample_name, num_samples):
        self.writer, self.sample_name, self.num_samples = writer, sample_name, num_samples
        self.loss_values = {k: 0.0 for k in self.loss_names}

    def add(self, values, num_samples):
        for loss in self.loss_names:
            self.loss_values[loss] += values[loss] * num_samples

    def push(self, epoch, reduce=None):
        """reduce: optional callable summing a list of floats over the ranks of a data-parallel job."""
        sums = [self.loss_values[k] for k in self.loss_names]
        if reduce is not None:
            sums = reduce(sums)
        means = {k: v / max(self.num_samples, 1) for k, v in zip(self.loss_names, sums)}
        if self.writer is not None:
            for k, v in means.items():
                self.writer.add_scalar(self.sample_name + '_' + k + '_loss', v, epoch)
        self.loss_values = {k: 0.0 for k in self.loss_names}
        return means


class PrecisionSummary:
    """utils.py:151-199: `<sample_name>_mAP` and `<sample_name>_AP_<label>` per epoch; nothing when no
    AP could be computed."""

    def __init__(self, writer, sample_name, labels):
        self.writer, self.sample_name, self.labels = writer, sample_name, labels

    def push(self, epoch, mAP, APs):
        if not APs or self.writer is None:
            return
        self.writer.add_scalar(self.sample_name + '_mAP', mAP, epoch)
        for label in self.labels:
            if label in APs:
                self.writer.add_scalar(self.sample_name + '_AP_' + label, APs[label], epoch)
