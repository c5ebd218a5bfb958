"""TEST INFRASTRUCTURE ONLY: CPU oracle (restatement + real-reference driver). Never imported by tengine_amd/."""
