"""Expert-demonstration data sets of the GAIL / imitation path: reader for the on-disk format the reference's
MimicExpWriter produces (USTC_lab/data/mimic_exp.py:17-140) and the batch order of the DataLoader the reference
wraps around it (GAIL.py:49-57).

Directory layout (written by the reference): ``dataset.txt`` -- first line = the directory, then one line per sample
``<frame file>,<frame file>,...||<label>``; frame files ``<proc>_<batch index>_<step>.npy`` hold ONE frame each, the
sample is their concatenation on axis 0 (mimic_exp.py:186-197).  Labels: atari = the action as text
(MimicExpClassificationReader, :206-212), classical / mujoco = a ``..._y.npy`` file (MimicExpRegressionReader, :215-232).
A sample whose files are missing is skipped, as the reference's try/except does.
"""
import os

import numpy as np
import torch


class MimicExpReader:
    regression = False

    def __init__(self, save_dir, module_type=torch.float32, device='cpu'):
        self.save_dir = save_dir
        with open(os.path.join(save_dir, "dataset.txt"), "r") as f:
            lines = [ln.strip() for ln in f.readlines()[1:]]
        self.data, self.line_list = {}, []
        self.dtype, self.device = module_type, device
        for line in lines:
            if "||" not in line:
                continue
            files, label = line.split("||")
            if self.regression:
                # MimicExpRegressionReader.to_memory (mimic_exp.py:219-229): a sample counts once, if ALL its files load
                try:
                    for name in files.split(","):
                        if self.data.get(name) is None:
                            self.data[name] = np.load(os.path.join(save_dir, name))[None]
                    self.data[label] = np.load(os.path.join(save_dir, label))
                except (OSError, ValueError):
                    continue
                self.line_list.append(line)
            else:
                # MimicExpAtariReader.to_memory (mimic_exp.py:240-251), quirks included: the writer stores only the FIRST frame
                # of every stacked sample, so a sample's later frames are the first frames of later steps; a line is appended
                # once per frame file it is the first to load, and a file that is missing when first asked for is marked and
                # never retried -- samples near the end of an episode therefore drop out and early ones repeat
                for name in files.split(","):
                    if self.data.get(name) is None:
                        try:
                            self.data[name] = np.load(os.path.join(save_dir, name))[None]   # [84, 84] -> [1, 84, 84]
                            self.line_list.append(line)
                        except (OSError, ValueError):
                            self.data[name] = -1

    def _label(self, label):
        if self.regression:
            return self.data[label]
        return np.array([int(float(label))], dtype=np.float32)

    def __getitem__(self, index):
        files, label = self.line_list[index].split("||")
        return np.concatenate([self.data[name] for name in files.split(",")], axis=0), self._label(label)

    def __len__(self):
        return len(self.line_list)


class MimicExpClassificationReader(MimicExpReader):
    regression = False


class MimicExpRegressionReader(MimicExpReader):
    regression = True


class MimicExpFactory:
    reader_register = {"atari": MimicExpClassificationReader, "mujoco": MimicExpRegressionReader,
                       "classical": MimicExpRegressionReader}

    def mimic_reader(self, task_type, *args):
        return self.reader_register[task_type](*args)


class batches:
    """DataLoader(dataset, batch_size, shuffle=True) as the discriminator consumes it: every ``iter()`` draws a fresh
    permutation the way torch's RandomSampler does (a seed from the global generator, then randperm on a private one), so
    that under the same ``torch.manual_seed`` the same samples form the batches; the last batch may be short."""

    def __init__(self, dataset, batch_size):
        self.dataset, self.batch_size = dataset, int(batch_size)

    def __len__(self):
        return (len(self.dataset) + self.batch_size - 1) // self.batch_size

    def __iter__(self):
        n = len(self.dataset)
        torch.empty((), dtype=torch.int64).random_()   # DataLoader's iterator draws its worker base seed first (_BaseDataLoaderIter)
        seed = int(torch.empty((), dtype=torch.int64).random_().item())
        g = torch.Generator()
        g.manual_seed(seed)
        order = torch.randperm(n, generator=g).tolist()
        for lo in range(0, n, self.batch_size):
            items = [self.dataset[i] for i in order[lo:lo + self.batch_size]]
            yield [torch.from_numpy(np.stack([x for x, _ in items])), torch.from_numpy(np.stack([y for _, y in items]))]
