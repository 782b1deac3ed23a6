"""Tokenised-text cache (second half): the reference tokenises every caption string on the host in every
step (/root/reference/run/train.py:66,72,74-75 -- `tokenizer(data['text'])`, 5 rephrases per clip plus the EgoClip narration) and
uploads the ids.  Captions repeat across epochs, so their token rows are kept in a device-resident table: a step tokenises only the
strings it has not seen, uploads those rows once, and gathers the batch's [len, context_length] id tensor on the device.

The tokenizer itself is out of scope: any callable `list[str] -> LongTensor [len, context_length]` works (the
reference's SimpleTokenizer, model/tokenizer.py).  Results are bit-identical to calling the tokenizer on the batch."""
import torch


class TokenizedTextCache:
    """Rows are stored as int32 (CLIP's vocabulary is 49 408 ids; 308 B per caption) and widened to int64 in the gather.  `max_rows`
    bounds the table (default 2^21 rows = 0.65 GB; EgoClip has ~3.8 M narrations x 5 rephrases, which unbounded would pin > 10 GB of
    HBM beside a step that sizes its batch to memory): once it is full, captions that are not in it are tokenised and uploaded for
    the call only (a bypass, no eviction -- the epoch's first 2 M distinct captions keep hitting)."""

    def __init__(self, tokenizer, context_length=77, device="cuda", capacity=1 << 16, max_rows=1 << 21):
        self.tokenizer, self.context_length, self.device = tokenizer, context_length, torch.device(device)
        self.index = {}                                               # caption -> row of the table
        self.max_rows = int(max_rows)
        self.table = torch.zeros((min(capacity, self.max_rows), context_length), dtype=torch.int32, device=self.device)
        self.hits = self.misses = self.bypassed = 0

    def __len__(self):
        return len(self.index)

    def __call__(self, texts):
        """list[str] -> int64 [len(texts), context_length] on the cache's device."""
        new = [t for t in dict.fromkeys(texts) if t not in self.index]
        extra = {}                                                    # captions beyond the cap: caption -> row of `rows_extra` (this call only)
        rows_extra = None
        if new:
            rows = self.tokenizer(new)
            if rows.shape != (len(new), self.context_length):
                raise ValueError("TokenizedTextCache: tokenizer returned %s for %d strings" % (tuple(rows.shape), len(new)))
            start = len(self.index)
            keep = max(0, min(len(new), self.max_rows - start))      # how many of the new captions still fit under the cap
            if keep and start + keep > self.table.shape[0]:           # grow geometrically up to the cap (a reallocation, not per step)
                grown = torch.zeros((min(self.max_rows, max(2 * self.table.shape[0], start + keep)), self.context_length), dtype=torch.int32, device=self.device)
                grown[:start] = self.table[:start]
                self.table = grown
            if keep:
                self.table[start:start + keep] = rows[:keep].to(device=self.device, dtype=torch.int32, non_blocking=True)
                for i, t in enumerate(new[:keep]):
                    self.index[t] = start + i
            if keep < len(new):
                rows_extra = rows[keep:].to(device=self.device, dtype=torch.int32, non_blocking=True)
                extra = {t: i for i, t in enumerate(new[keep:])}
                self.bypassed += len(new) - keep
        self.misses += len(new)
        self.hits += len(texts) - len(new)
        if rows_extra is None:
            return self.table.index_select(0, self._dev_index([self.index[t] for t in texts])).to(torch.int64)
        # Some captions are not in the (full) table: gather the cached rows from the table and the bypassed rows from this call's
        # upload, each into its own positions of the output.  The table itself is never copied (concatenating it with
        # `rows_extra` moved 0.65 GB per step once the cap was reached).
        pos_c = [i for i, t in enumerate(texts) if t in self.index]
        pos_x = [i for i, t in enumerate(texts) if t not in self.index]
        out = torch.empty((len(texts), self.context_length), dtype=torch.int64, device=self.device)
        if pos_c:
            out[self._dev_index(pos_c)] = self.table.index_select(0, self._dev_index([self.index[texts[i]] for i in pos_c])).to(torch.int64)
        out[self._dev_index(pos_x)] = rows_extra.index_select(0, self._dev_index([extra[texts[i]] for i in pos_x])).to(torch.int64)
        return out

    def _dev_index(self, rows):
        idx = torch.tensor(rows, dtype=torch.int64)
        if self.device.type == "cuda":
            idx = idx.pin_memory().to(self.device, non_blocking=True)
        return idx
