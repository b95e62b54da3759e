"""Host -> device input pipeline for page batches.

The reference decodes PNG layers to float64 `(1,H,W,C)/255` arrays per iteration and moves them
with `CP.copy` on the compute stream (my_model/datasets.py:113-124, train_data_generator.py:24-37,
model.py:413-416).  Here a batch travels as uint8 (what the PNGs hold: 4x fewer PCIe bytes than
float32, 8x fewer than the reference's float64) through pinned double buffers on a dedicated copy
stream, and becomes float on the device (`uocr_u8_to_float`, scale 1/255 for the image, 1 for the
0/1 label layers) -- so the upload of batch i+1 overlaps the train step of batch i.
"""
import numpy as np
import torch

from ..nn import ops
from ..nn.gpu import CP, DeviceArray

# context label -> (layer tag, scale): the TRAIN_PAGE mapping of trainer.PageTrainer.make_context
PAGE_FEEDS = {'monochrome_X': ('image', 1.0 / 255.0), 'monochrome_y': ('monochrome', 1.0),
              'paragraph_X': ('monochrome', 1.0), 'paragraph_y': ('paragraph', 1.0),
              'line_X': ('monochrome', 1.0), 'line_y': ('line', 1.0),
              'char_X': ('char_lines', 1.0 / 255.0), 'char_y': ('char_labels', 1.0)}


def to_uint8_layers(layers):
    """Float layers of synthetic.make_page_batch -> the uint8 form a PNG dataset delivers."""
    out = {}
    for tag, arr in layers.items():
        scale = 255.0 if tag in ('image', 'char_lines') else 1.0
        out[tag] = np.clip(np.rint(arr * scale), 0, 255).astype(np.uint8)
    return out


class PageFeeder:
    def __init__(self, example_layers_u8, feeds=None, slots=2):
        self.feeds = dict(PAGE_FEEDS if feeds is None else feeds)
        self.tags = sorted({tag for tag, _ in self.feeds.values()})
        rt = CP.runtime()
        self.copy_stream = torch.cuda.Stream(device=rt.device)
        self.slots = []
        for _ in range(slots):
            host = {t: torch.empty(example_layers_u8[t].shape, dtype=torch.uint8).pin_memory() for t in self.tags}
            dev = {t: torch.empty(example_layers_u8[t].shape, dtype=torch.uint8, device=rt.device) for t in self.tags}
            self.slots.append({'host': host, 'dev': dev, 'ready': torch.cuda.Event(), 'free': torch.cuda.Event()})
        self._staged = []
        self._next = 0

    def stage(self, layers_u8):
        """Copy a host batch into the next pinned slot and start its upload on the copy stream."""
        slot = self.slots[self._next]
        self._next = (self._next + 1) % len(self.slots)
        slot['free'].synchronize()                    # the step that read this slot's device buffers is done
        for t in self.tags:
            slot['host'][t].numpy()[...] = layers_u8[t]
        with torch.cuda.stream(self.copy_stream):
            for t in self.tags:
                slot['dev'][t].copy_(slot['host'][t], non_blocking=True)
            slot['ready'].record(self.copy_stream)
        self._staged.append(slot)

    def context(self, into=None):
        """Device float context of the oldest staged batch; the compute stream waits for its upload only.
        `into` = {label: DeviceArray}: convert into these arrays (the static inputs of a graph-replaying
        trainer, `PageTrainer.static_inputs()`) instead of allocating; the caller makes sure nothing still
        reads them (`PageTrainer.join()`)."""
        slot = self._staged.pop(0)
        compute = torch.cuda.current_stream()
        compute.wait_event(slot['ready'])
        made, context = {}, {}
        for label, (tag, scale) in self.feeds.items():
            target = None if into is None else into.get(label)      # labels of eager nets: a fresh array
            key = (tag, scale) if target is None else (tag, scale, id(target))
            if key not in made:
                made[key] = ops.u8_to_float(DeviceArray(slot['dev'][tag]), scale, out=target)
            context[label] = made[key]
        slot['free'].record(compute)
        return context
