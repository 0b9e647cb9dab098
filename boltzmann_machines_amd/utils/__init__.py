from .rng import RNG
from .utils import (batch_iter, epoch_iter, make_list_from, write_during_training,
                    log_sum_exp, log_mean_exp, log_diff_exp, log_std_exp)
