"""Meters for the step loop -- counterpart of /root/reference/logs/logger.py:27-161 (SmoothedValue / MetricLogger).

Same contract as the reference (``update(name=value)``, ``meters[name].global_avg`` = total / count,
``log_every(iterable, print_freq, header, rank)``), with one MI355X-motivated difference: values may be 0-dim DEVICE
tensors.  They are accumulated on the device and only read back when a line is printed or ``global_avg`` is taken, so the
hot loop has no per-step ``.item()`` host sync (the reference does three per step: tools/engine.py:71-73).
"""
import datetime
import time
from collections import defaultdict, deque

import torch


class SmoothedValue:
    def __init__(self, window_size=20, fmt=None):
        self.fmt = fmt or "{median:.4f} ({global_avg:.4f})"
        self.deque = deque(maxlen=window_size)
        self.total = 0.0
        self.count = 0
        self._dev_total = None

    def update(self, value, n=1, defer=None):
        """``defer``: a list that collects (device total, addend) pairs instead of adding here -- MetricLogger.update adds the
        meters of one call in a single multi-tensor launch."""
        self.deque.append(value)
        self.count += n
        if isinstance(value, torch.Tensor):
            v = value.detach()
            if v.dtype != torch.float32:
                v = v.float()
            if n != 1:
                v = v * n
            if self._dev_total is None:
                self._dev_total = v.clone()
            elif defer is not None:
                defer.append((self._dev_total, v))
            else:
                self._dev_total.add_(v)
        else:
            self.total += value * n

    def _window(self):
        return [v.item() if isinstance(v, torch.Tensor) else v for v in self.deque]

    @property
    def median(self):
        return torch.tensor(self._window()).median().item()

    @property
    def avg(self):
        return torch.tensor(self._window(), dtype=torch.float32).mean().item()

    @property
    def global_avg(self):
        total = self.total + (self._dev_total.item() if self._dev_total is not None else 0.0)
        return total / max(self.count, 1)

    @property
    def value(self):
        v = self.deque[-1]
        return v.item() if isinstance(v, torch.Tensor) else v

    def __str__(self):
        return self.fmt.format(median=self.median, avg=self.avg, global_avg=self.global_avg, value=self.value)


class MetricLogger:
    def __init__(self, delimiter="  ", printer=print):
        self.meters = defaultdict(SmoothedValue)
        self.delimiter = delimiter
        self.printer = printer

    def update(self, **kwargs):
        pend = []
        for k, v in kwargs.items():
            if v is None:
                continue
            assert isinstance(v, (float, int, torch.Tensor))
            self.meters[k].update(v, defer=pend)
        if len(pend) == 1:
            pend[0][0].add_(pend[0][1])
        elif pend:
            by_dev = {}
            for t, v in pend:
                by_dev.setdefault((t.device, v.shape == t.shape), []).append((t, v))
            for (_, same), pairs in by_dev.items():
                if same:
                    torch._foreach_add_([t for t, _ in pairs], [v for _, v in pairs])       # one launch for all meters of this call
                else:
                    for t, v in pairs:
                        t.add_(v)

    def __getattr__(self, attr):
        if attr in self.meters:
            return self.meters[attr]
        raise AttributeError(attr)

    def __str__(self):
        return self.delimiter.join(f"{name}: {meter}" for name, meter in self.meters.items())

    def log_every(self, iterable, print_freq, header=None, rank=0):
        header = header or ""
        start = time.time()
        n = len(iterable) if hasattr(iterable, "__len__") else None
        for i, obj in enumerate(iterable):
            yield obj
            if rank == 0 and print_freq and (i % print_freq == 0 or (n is not None and i == n - 1)):
                el = time.time() - start
                eta = "" if n is None else f"eta: {datetime.timedelta(seconds=int(el / (i + 1) * (n - i - 1)))}"
                self.printer(self.delimiter.join([header, f"[{i}/{n if n is not None else '?'}]", eta, str(self),
                                                  f"time: {el / (i + 1):.4f}"]))
        if rank == 0:
            total = time.time() - start
            self.printer(f"{header} Total time: {datetime.timedelta(seconds=int(total))}")
