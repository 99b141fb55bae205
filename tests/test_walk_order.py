"""The order locate_pass_kernel walks its batches in (multimesh_amd/csrc/mm_locate_hex8.hip, "Order of the fresh batches"):
the kernel's sequence -> batch arithmetic, restated here line by line, must visit every batch of an XCD's range exactly
once for any range length, plane count and panel width -- correctness of the pass rests on nothing else about it."""
import random


def walk(nb_x, planes, nbatches, panel):
    rows = (planes * nb_x + nbatches // 2) // nbatches
    rows = max(rows, 1)
    cols = (nb_x + rows - 1) // rows
    seq_end = (cols + panel - 1) // panel * panel * rows
    out = []
    for seq in range(seq_end):
        b = seq
        if rows > 1:
            per_panel = rows * panel
            pnl, rem = divmod(seq, per_panel)
            row = rem // panel
            col = pnl * panel + (rem - row * panel)
            if col >= cols:
                continue
            b = row * cols + col
        if b < nb_x:
            out.append(b)
    return out


def test_every_batch_once():
    rng = random.Random(11)
    for _ in range(400):
        nbatches = rng.randint(1, 200_000)
        xcd = rng.randrange(8)
        nb_x = nbatches * (xcd + 1) // 8 - nbatches * xcd // 8
        planes = rng.choice([1, 2, 7, 54, 108, 233, 1024])
        panel = rng.choice([1, 3, 32, 64, 256, 257, 4096])
        assert sorted(walk(nb_x, planes, nbatches, panel)) == list(range(nb_x)), (nb_x, planes, nbatches, panel)


def test_a_wave_moves_from_plane_to_plane():
    # the metric mesh: 157,464 batches, 108 planes, one XCD's eighth, 256 waves = the panel
    nb_x, panel = 19683, 256
    order = walk(nb_x, 108, 157464, panel)
    rows = (108 * nb_x + 157464 // 2) // 157464
    cols = (nb_x + rows - 1) // rows
    w = 17                                     # wave 17's first batches: sequence numbers 17, 17 + 256, ...
    mine = [order[w + k * panel] for k in range(rows)]
    assert [b % cols for b in mine] == [w] * rows          # the same column ...
    assert [b // cols for b in mine] == list(range(rows))  # ... one plane on each time
