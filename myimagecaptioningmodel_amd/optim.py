"""Host side of the optimizer: Paddle-form Adam step bookkeeping and the learning-rate value.

`get_lr` restates /root/reference/ImageCaptioning/tools/util.py:20-119 as plain host scalars:
the reference builds these schedules out of graph ops evaluated once per step; their inputs
are only the global step counter (`@LR_DECAY_COUNTER@`, util.py:47-51) and, for
`cosine_decay_restart_warmup`, a persistable epoch counter (util.py:94-101).
"""
import math

ADAM_BETA1, ADAM_BETA2, ADAM_EPS = 0.9, 0.999, 1e-8      # fluid.optimizer.Adam defaults (IC/train.py:31)

STRATEGIES = (None, 'cosine_decay', 'cosine_decay_restart', 'cosine_decay_restart_warmup', 'cosine_decay_warmup')


class LRSchedule:
    def __init__(self, strategy, base_lr, sample_cnt, batch_size, decay_epoch=0, warmup_epoch=3, max_epoch=10):
        if strategy not in STRATEGIES:
            raise ValueError('Lr衰减策略错误')                     # util.py:21-23
        self.strategy, self.base_lr = strategy, float(base_lr)
        self.step_each_epoch = max(1, math.ceil(sample_cnt / batch_size)) if sample_cnt else 1   # util.py:24
        self.decay_epoch, self.warmup_epoch, self.max_epoch = decay_epoch, warmup_epoch, max_epoch

    # ---- the two persistable scalars behind the schedules, as pure functions of the number of steps taken
    def counter_begin(self):
        """`begin` of `_decay_step_counter` for this strategy (util.py:47-51,55,98: the warm-up forms count from 1;
        fluid.layers.cosine_decay and cosine_decay_restart from 0)."""
        return 1 if self.strategy in ('cosine_decay_warmup', 'cosine_decay_restart_warmup') else 0

    def counter_after(self, steps_taken):
        """Value held by `@LR_DECAY_COUNTER@` after `steps_taken` runs of the train program: the variable starts at
        begin - 1 and is incremented inside every run (Paddle autoincreased_step_counter, from memory)."""
        return self.counter_begin() - 1 + int(steps_taken)

    def steps_from_counter(self, counter):
        return max(0, int(counter) - self.counter_begin() + 1)

    def cur_epoch_after(self, steps_taken):
        """The persistable `cur_epoch` of util.py:94-101 after `steps_taken` runs: it starts at 0 and gains 1 in
        every run whose counter value g (= 1, 2, ...) satisfies g % step_each_epoch == 0."""
        return float(int(steps_taken) // self.step_each_epoch)

    @staticmethod
    def _restart_fraction(completed_fraction, t_mul=2.0):
        """util.py:77-84 / :105-110 (t_mul = 2.0, m_mul = 1.0)."""
        i_restart = math.floor(math.log(1.0 - completed_fraction * (1.0 - t_mul)) / math.log(t_mul))
        sum_r = (1.0 - t_mul ** i_restart) / (1.0 - t_mul)
        return (completed_fraction - sum_r) / t_mul ** i_restart

    def value(self, step):
        """lr used by training step number `step` (0-based count of steps already taken).  Pure: calling it twice, or
        out of order, changes nothing (the reference keeps `cur_epoch` as a persistable; here it is derived)."""
        s, lr = self.strategy, self.base_lr
        if s is None:
            return lr                                                            # util.py:43-44
        if s == 'cosine_decay':                                                  # fluid.layers.cosine_decay, counter from 0
            epoch = math.floor(step / self.step_each_epoch)
            return lr * 0.5 * (math.cos(epoch * math.pi / self.decay_epoch) + 1)
        if s == 'cosine_decay_warmup':                                           # util.py:54-67, counter from 1
            start_lr = 0.00001
            cur_epoch = math.floor((step + 1) / self.step_each_epoch)
            if cur_epoch < self.warmup_epoch:
                return start_lr + (lr - start_lr) / self.warmup_epoch * cur_epoch
            return 0.5 * lr * (math.cos((cur_epoch - self.warmup_epoch) * math.pi / float(self.max_epoch - self.warmup_epoch)) + 1)
        if s == 'cosine_decay_restart':                                          # util.py:70-89, counter from 0
            cur_epoch = math.floor(step / self.step_each_epoch)
            frac = self._restart_fraction(cur_epoch / self.decay_epoch)
            return lr * 0.5 * (math.cos(math.pi * frac) + 1)
        # cosine_decay_restart_warmup, util.py:92-119: counter from 1; cur_epoch += 1 in the run whose counter g = step + 1
        # is a multiple of step_each_epoch, BEFORE the lr of that run is formed (:99-101) -- so run `step` sees
        # cur_epoch = floor((step + 1) / spe): a pure function of the step, like the other strategies
        start_lr = 0.00001
        cur_epoch = self.cur_epoch_after(step + 1)
        if cur_epoch < self.warmup_epoch:
            return start_lr + (lr - start_lr) * (cur_epoch / float(self.warmup_epoch))
        frac = self._restart_fraction((cur_epoch - self.warmup_epoch) / self.decay_epoch)
        return lr * 0.5 * (math.cos(math.pi * frac) + 1)


def adam_lr_t(lr, step):
    """Bias-corrected step size of Paddle-1.8's adam op, `step` counted from 1."""
    return lr * math.sqrt(1.0 - ADAM_BETA2 ** step) / (1.0 - ADAM_BETA1 ** step)
