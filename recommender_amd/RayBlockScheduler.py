"""Host mirror of the reference's free-block scheduler (RayBlockScheduler.py:14-116) and worker loop
(:148-181).  Pure control logic, no arithmetic: on the throughput path the same policy runs ON the GPU
inside the epoch kernel (csrc/epoch.hip); this class exists so that code written against the reference's
scheduler API keeps working and so that the policy itself is testable against traces captured from the
reference (tests/golden/scheduler.npz).

Tie-break: deterministic=False draws with random.choice exactly like RayBlockScheduler.py:58 (so a
random.seed() reproduces the reference's sequence); deterministic=True takes the first strict minimum
in scan order like the multiprocessing prototype (MPBlockScheduler.py:48-54)."""
import random

from .numba_methods import block_sgd as sgd  # the block-local kernel lives next to the scheduler, as in the reference


class BlockScheduler:
    def __init__(self, width, iters, verbose=False, deterministic=False):
        self.width = width
        self.iters = iters
        self.verbose = verbose
        self.deterministic = deterministic
        self.n_threads = width - 1
        self.update_counter = [[0] * width for _ in range(width)]
        self.unused_rows = list(range(width))
        self.unused_cols = list(range(width))
        self.completed_rows, self.completed_cols = set(), set()
        self.test = 0

    def check_completion(self):
        return len(self.completed_rows) == self.width and len(self.completed_cols) == self.width

    def get_next(self, completed=None):
        """Release `completed` (if any) and hand out a block whose row and column are both free and whose
        update count is minimal and below `iters`; None when there is none right now."""
        if completed:
            self.completed_chunk(completed)
        if not self.unused_cols:
            return None
        best, ties = None, []
        for i in self.unused_rows:
            row = self.update_counter[i]
            for j in self.unused_cols:
                v = row[j]
                if v >= self.iters:
                    continue
                if best is None or v < best:
                    best, ties = v, [(i, j)]
                elif v == best and not self.deterministic:
                    ties.append((i, j))
        if not ties:
            return None
        pick = ties[0] if self.deterministic else random.choice(ties)
        self.unused_rows.remove(pick[0])
        self.unused_cols.remove(pick[1])
        if self.verbose:
            print("Chose", pick)
        return pick

    def completed_chunk(self, idx):
        row, col = idx
        self.test += 1
        if row in self.unused_rows or col in self.unused_cols:
            raise Exception("Desync occurred, completed chunk was still in unused chunks")
        self.update_counter[row][col] += 1
        row_full = all(v == self.iters for v in self.update_counter[row])
        col_full = all(self.update_counter[i][col] == self.iters for i in range(self.width))
        if col_full:
            self.completed_cols.add(col)
        else:
            self.unused_cols.append(col)
        if row_full:
            self.completed_rows.add(row)
        else:
            self.unused_rows.append(row)

    def get_update_counter(self):
        return self.update_counter

    def get_test(self):
        return self.test


def consumer(scheduler, trainer, groups, row_ptrs, col_inds):
    """The reference's worker loop (RayBlockScheduler.py:149-181) as a plain function: one caller drains
    the scheduler, running every block through the HIP block kernel.  With the M2 defect repaired: the
    slices are views of the trainer's arrays, so updates land in place (y included)."""
    block = scheduler.get_next()
    count = 0
    while block is not None:
        ss = trainer.make_subsample(block)
        sgd(ss.idx_offset, ss.P, ss.Q, ss.b_u, ss.b_i, ss.b, ss.y if ss.y is not False else None,
            groups[block[0]][block[1]], row_ptrs, col_inds, ss.alpha, ss.beta1, ss.beta2)
        trainer.update_params(ss)
        trainer.increment()
        count += 1
        block = scheduler.get_next(completed=block)
    return count
