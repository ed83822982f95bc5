#!/bin/bash
timeout 900 python tools/chunk_ab.py
