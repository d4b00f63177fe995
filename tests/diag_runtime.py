import sys, os
sys.path.insert(0, os.getcwd())
mode = sys.argv[1]
from yadcc_amd import binding
if mode == "ydc_first_torch_import":
    binding.lib(); import torch; print("count", binding.device_count(), binding.lib().ydc_last_error(None))
elif mode == "ydc_first_torch_cuda":
    binding.lib(); import torch; print(torch.cuda.is_available()); print("count", binding.device_count(), binding.lib().ydc_last_error(None))
elif mode == "torch_cuda_first":
    import torch; print(torch.cuda.is_available()); print("count", binding.device_count(), binding.lib().ydc_last_error(None))
elif mode == "ydc_init_then_torch_cuda":
    print("count", binding.device_count()); import torch; print(torch.cuda.is_available(), torch.zeros(3, device="cuda").sum().item()); c = binding.Context(); print("ctx ok")
