"""SGDR cosine schedule with growing period and intra-epoch updates (host-side scalar math).

Counterpart of the reference's train_test_code/warm_restarts_lr.py:14-63: same constructor, ``intra_epoch_step(ratio)``,
``step()``, ``just_restarted`` and state_dict fields, so checkpoints written by either side load in the other.
"""
import math

from torch.optim.lr_scheduler import _LRScheduler

__all__ = ['WarmRestartLR']


class WarmRestartLR(_LRScheduler):
    def __init__(self, optimizer, init_run_period_epochs=10, lr_min=0, last_epoch=-1, growth_factor=2):
        self.cur_run_period_epochs = init_run_period_epochs
        self.lr_min = lr_min
        self.next_restart_epoch = init_run_period_epochs
        self.last_restart_epoch = max(last_epoch, 0)
        self.period_growth_factor = growth_factor
        self.cur_epoch_ratio = 0
        self.just_restarted = False
        super().__init__(optimizer, last_epoch)

    def _cosine(self):
        pos = (self.last_epoch - self.last_restart_epoch + self.cur_epoch_ratio) / self.cur_run_period_epochs
        return 1 + math.cos(math.pi * pos)

    def get_lr(self):
        assert -1.0e-12 < self.cur_epoch_ratio < 1 + 1.0e-12
        c = self._cosine()
        return [self.lr_min + (base - self.lr_min) / 2 * c for base in self.base_lrs]

    def intra_epoch_step(self, epoch_ratio):
        """Set the LR for a fractional position inside the current epoch (train.py:427-428)."""
        self.cur_epoch_ratio = epoch_ratio
        for group, lr in zip(self.optimizer.param_groups, self.get_lr()):
            group['lr'] = lr

    def step(self, epoch=None):
        self.cur_epoch_ratio = 0
        super().step(epoch)
        self.just_restarted = self.last_epoch >= self.next_restart_epoch
        if self.just_restarted:
            print('WARM RESTART AFTER PERIOD OF {} EPOCHS'.format(self.cur_run_period_epochs))
            self.last_restart_epoch = self.next_restart_epoch
            self.cur_run_period_epochs *= self.period_growth_factor
            self.next_restart_epoch += self.cur_run_period_epochs
