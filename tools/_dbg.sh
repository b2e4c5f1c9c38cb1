for v in "S2SVC_NO_BRANCH=1" "X=1"; do echo "=== $v"; env $v python tools/_dbg_stage.py 2>&1 | grep "replay\|eager\|differs\|Error\|error" | head -40; done
timeout 900 python -m pytest tests/test_gpu_models.py -m gpu -q -k "memory_cut or dp_ or fs2vc" 2>&1 | tail -15
