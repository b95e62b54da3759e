"""Seeded synthetic page batches (the reference's generate_data needs Windows TTF fonts and Faker,
image_generator/generate.py, primitives/__init__.py:100-110 -- unavailable here).

A page is white (1.0) with dark ink strokes (axis-aligned rectangles) laid out as paragraphs ->
lines -> glyph strokes, ~5-15 % ink coverage; the label layers follow the reference's layer tags
(my_model/constants.py:9-29):
    image       (B,H,W,1) float in [0,1]
    monochrome  (B,H,W,1) {0,1}: ink mask (image < 0.5)
    paragraph   (B,H,W,1) {0,1}: filled paragraph boxes
    line        (B,H,W,2) {0,1}: line_top / line_bottom bands
    char_lines  (Bc,32,Wc,1) float: line strips for the Char net, char_labels (Bc*Wc,162) one-hot
"""
import numpy as np

from .model import CHAR_INPUT_HEIGHT, N_CHARS


def make_page_batch(batch, height=256, width=512, char_width=64, seed=1234, char_batch=None):
    rng = np.random.default_rng(seed)
    image = np.ones((batch, height, width, 1))
    paragraph = np.zeros((batch, height, width, 1))
    line = np.zeros((batch, height, width, 2))
    for b in range(batch):
        y = int(rng.integers(4, 12))
        while y < height - 24:
            par_h = int(rng.integers(24, max(25, min(96, height - y))))
            x0 = int(rng.integers(4, max(5, width // 8)))
            x1 = int(rng.integers(width * 5 // 8, width - 4))
            paragraph[b, y:y + par_h, x0:x1, 0] = 1
            ly = y + 2
            while ly + 10 <= y + par_h:
                lh = int(rng.integers(8, 13))
                if ly + lh > y + par_h:
                    break
                line[b, ly:ly + 2, x0:x1, 0] = 1
                line[b, ly + lh - 2:ly + lh, x0:x1, 1] = 1
                cx = x0 + 2
                while cx < x1 - 8:
                    cw = int(rng.integers(3, 8))
                    if rng.random() < 0.85:
                        sy = ly + 2 + int(rng.integers(0, 2))
                        image[b, sy:ly + lh - 2, cx:cx + max(1, cw // 3), 0] = rng.uniform(0.0, 0.25)
                        if rng.random() < 0.6:
                            my = int(rng.integers(sy, ly + lh - 2))
                            image[b, my:my + 1, cx:cx + cw, 0] = rng.uniform(0.0, 0.25)
                    cx += cw + int(rng.integers(1, 4))
                ly += lh + int(rng.integers(2, 5))
            y += par_h + int(rng.integers(8, 20))
    image = np.clip(image + rng.normal(0, 0.02, image.shape), 0, 1)
    monochrome = (image < 0.5).astype(np.float64)
    cb = batch if char_batch is None else char_batch
    char_lines = np.ones((cb, CHAR_INPUT_HEIGHT, char_width, 1))
    labels = rng.integers(0, N_CHARS, (cb, char_width))
    for b in range(cb):
        for x in range(0, char_width - 4, 6):
            h0 = int(rng.integers(4, 12))
            char_lines[b, h0:h0 + int(rng.integers(8, 18)), x:x + int(rng.integers(1, 4)), 0] = rng.uniform(0, 0.3)
    char_labels = np.zeros((cb * char_width, N_CHARS))
    char_labels[np.arange(cb * char_width), labels.reshape(-1)] = 1
    return {'image': image, 'monochrome': monochrome, 'paragraph': paragraph, 'line': line,
            'char_lines': char_lines, 'char_labels': char_labels}


class SyntheticPages:
    """Dataset-like source with the reference's `get(index, layer_tags=...)` call shape
    (my_model/datasets.py:88-161) over seeded synthetic batches."""

    def __init__(self, batch, height=256, width=512, char_width=64, seed=1234, length=4):
        self.args = (batch, height, width, char_width)
        self.seed, self.length = seed, length
        self._cache = {}

    def __len__(self):
        return self.length

    def get(self, index, layer_tags=None):
        if index not in self._cache:
            self._cache[index] = make_page_batch(*self.args, seed=self.seed + index)
        layers = self._cache[index]
        return layers if layer_tags is None else {tag: layers[tag] for tag in layer_tags}
