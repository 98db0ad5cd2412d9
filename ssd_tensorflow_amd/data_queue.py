"""Shared-memory slot ring between the feeder's worker processes and the training process: the
counterpart of the reference's DataQueue (data_queue.py:26-112) and of the worker set-up of
TrainingData.gen_batch (training_data.py:147-195).

What the reference moves through its slots is the finished float32 batch (images [b,H,W,3] + labels
[b,A,C+5]: 62 MB at batch 32) which the parent copies out and feeds to ONE device.  Here a batch is born
in HBM, so what a worker produces is the batch's *recipe*: the source images' uint8 bytes, one small
parameter record per image and the transformed ground-truth boxes (transforms.plan_params).  A slot is a
flat byte buffer holding a few named numpy arrays back to back; the parent uploads it to the GPU where the
augmentation and label kernels turn it into the batch.

Differences from the reference, on purpose:
  * slots are handed out by the PARENT together with the task (not grabbed by whichever worker finishes
    first), and the parent consumes results in task order: batches come out in the order of the sample
    list whatever the number of workers (the reference yields in completion order, which makes an epoch
    irreproducible as soon as num_workers > 1);
  * a payload that does not fit its slot travels through the result pipe instead of raising;
  * put() keeps the reference's argument checks (data_queue.py:63-79): anything but C-contiguous numpy
    arrays is a ValueError.
"""
import multiprocessing as mp
import queue as q

import numpy as np

ALIGN = 256


def _aligned(n):
    return (n + ALIGN - 1) // ALIGN * ALIGN


class WorkerError(RuntimeError):
    """A task failed in its worker process; carries the task's tag and slot so the parent can take the slot back."""

    def __init__(self, tag, arr_id, message):
        super().__init__('feeder worker failed:\n' + str(message))
        self.tag, self.arr_id = tag, arr_id


class DataQueue:
    """maxsize slots of slot_bytes each, created before the workers start (shared memory, like the reference's
    mp.Array('c', n, lock=False) slots)."""

    def __init__(self, slot_bytes, maxsize, ctx=None):
        ctx = ctx or mp.get_context('fork')
        self.slot_bytes = int(_aligned(max(int(slot_bytes), ALIGN)))
        self.maxsize = int(maxsize)
        self._buffers = [ctx.RawArray('c', self.slot_bytes) for _ in range(self.maxsize)]
        self.array_pool = [np.frombuffer(b, dtype=np.uint8) for b in self._buffers]
        self.queue = ctx.Queue()

    # ---- travelling to a worker started through the fork server: the slots by handle, the views rebuilt over them -----------
    def __getstate__(self):
        st = dict(self.__dict__)
        st['array_pool'] = None
        return st

    def __setstate__(self, st):
        self.__dict__.update(st)
        self.array_pool = [np.frombuffer(b, dtype=np.uint8) for b in self._buffers]

    # ---- worker side ------------------------------------------------------------------------------------------
    def put(self, tag, arr_id, arrays, boxes):
        """Store `arrays` ({name: ndarray}) in slot arr_id and announce (tag, arr_id, layout, boxes).  A set of
        arrays larger than the slot is sent through the pipe (layout None, the arrays themselves attached)."""
        layout, off = [], 0
        for name, a in arrays.items():
            if type(a) is not np.ndarray:
                raise ValueError(name + ' needs to be a numpy array')                          # data_queue.py:64-65
            if not a.flags['C_CONTIGUOUS']:
                raise ValueError(name + ' needs to be C-contiguous')
            layout.append((name, a.dtype.str, a.shape, off, a.nbytes))
            off = _aligned(off + a.nbytes)
        if off > self.slot_bytes:
            self.queue.put((tag, arr_id, None, boxes, {k: np.ascontiguousarray(v) for k, v in arrays.items()}))
            return
        slot = self.array_pool[arr_id]
        for (name, _, _, o, nb), a in zip(layout, arrays.values()):
            slot[o:o + nb] = a.reshape(-1).view(np.uint8)
        self.queue.put((tag, arr_id, layout, boxes, None))

    def put_error(self, tag, arr_id, message):
        self.queue.put((tag, arr_id, 'error', message, None))

    # ---- parent side -------------------------------------------------------------------------------------------
    def get(self, *args, **kwargs):
        """-> (tag, arr_id, arrays, boxes): `arrays` are VIEWS into the slot (no copy: the parent uploads them
        and only then gives the slot to the next task); raises queue.Empty on timeout."""
        tag, arr_id, layout, boxes, attached = self.queue.get(*args, **kwargs)
        if isinstance(layout, str) and layout == 'error':
            raise WorkerError(tag, arr_id, boxes)
        if layout is None:
            return tag, arr_id, attached, boxes
        slot = self.array_pool[arr_id]
        arrays = {name: slot[o:o + nb].view(np.dtype(dt)).reshape(shape) for name, dt, shape, o, nb in layout}
        return tag, arr_id, arrays, boxes

    def empty(self):
        return self.queue.empty()


Empty = q.Empty
