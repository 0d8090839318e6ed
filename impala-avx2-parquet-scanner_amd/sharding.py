"""Row-stripe sharding of a column chunk across the GPUs of one node + all-gather of the
per-stripe selection bitmaps.

Blocks of 64 rows are independent ("Row groups are independent, so this could be parallelized",
hdfs-parquet-scanner.cc:1056-1060), so rank r owns rows [r*S, (r+1)*S) with S a multiple of the
2048-row sub-tile: every stripe starts on an FLE block, a bitmap word and a batch boundary, for
every column of the predicate.  There is no data-path collective besides the one exchange of
bitmap words: torch.distributed.all_gather_into_tensor, which is RCCL over xGMI with the "nccl"
backend (and gloo in the CPU tests).
"""
import torch
import torch.distributed as dist

ALIGN_ROWS = 2048


def stripe_rows(n_rows, world):
    """Rows per stripe: equal for all ranks, multiple of ALIGN_ROWS."""
    tiles = (n_rows + ALIGN_ROWS - 1) // ALIGN_ROWS
    return ((tiles + world - 1) // world) * ALIGN_ROWS


def stripe_bounds(n_rows, world, rank):
    """[row0, row1) of rank's stripe (row1 == row0 for ranks beyond the data)."""
    s = stripe_rows(n_rows, world)
    row0 = min(rank * s, n_rows)
    return row0, min(row0 + s, n_rows)


def cyclic_pieces(n_rows, world, rank, n_chunks):
    """Block-cyclic stripes for an exchange that overlaps the scan INSIDE one step (SURVEY 8e:
    "chunk the stripe ... overlap gather of chunk i with scan of chunk i+1").

    The column is cut into n_chunks * world pieces of piece_rows rows (a multiple of ALIGN_ROWS);
    rank r owns pieces r, world + r, 2*world + r, ...  All ranks' i-th pieces are adjacent in the
    column, so all-gathering the bitmap words of chunk i (piece_rows / 64 words per rank) fills
    words [i*world*pw, (i+1)*world*pw) of the global bitmap in natural row order: the gathered
    bitmap is bit-identical to a single-GPU scan and chunk i can be on the wire while chunk i+1
    is still being scanned.  Returns (piece_rows, [(row0, row1)] * n_chunks); pieces beyond the
    data are empty (row0 == row1)."""
    piece_rows = stripe_rows(n_rows, world * n_chunks)
    pieces = []
    for i in range(n_chunks):
        row0 = min((i * world + rank) * piece_rows, n_rows)
        pieces.append((row0, min(row0 + piece_rows, n_rows)))
    return piece_rows, pieces


def allgather_chunk(local_words, full_words, chunk, piece_words, world, group=None):
    """All-gather chunk `chunk` of every rank into its place of the global bitmap (torch.distributed
    form of ips_allgather_bitmap; used by the CPU tests with gloo).  local_words: this rank's
    piece_words words of the chunk (zero-padded when the piece is short or empty)."""
    send = local_words
    if send.numel() != piece_words:
        send = torch.zeros(piece_words, dtype=local_words.dtype, device=local_words.device)
        send[:local_words.numel()] = local_words
    dst = full_words[chunk * world * piece_words:(chunk + 1) * world * piece_words]
    return dist.all_gather_into_tensor(dst, send.contiguous(), group=group, async_op=True)


def stripe_word_slice(enc_words, bit_width, n_rows, world, rank):
    """The encoded words (FLE blocks) of this rank's stripe: w words per 64-row block."""
    row0, row1 = stripe_bounds(n_rows, world, rank)
    return enc_words[(row0 // 64) * bit_width: ((row1 + 63) // 64) * bit_width]


def allgather_bitmap(local_words, n_rows, world, group=None):
    """local_words: this stripe's bitmap words (ceil(stripe rows / 64), int64 tensor on the
    device of the backend).  Returns the whole column's bitmap, ceil(n_rows/64) words, on every
    rank -- bit-identical to a single-GPU scan."""
    s_words = stripe_rows(n_rows, world) // 64
    send = local_words
    if send.numel() != s_words:  # short last stripe / empty stripe: pad with zero words
        send = torch.zeros(s_words, dtype=local_words.dtype, device=local_words.device)
        send[:local_words.numel()] = local_words
    full = torch.empty(s_words * world, dtype=local_words.dtype, device=local_words.device)
    dist.all_gather_into_tensor(full, send.contiguous(), group=group)
    return full[:(n_rows + 63) // 64]
