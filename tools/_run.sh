#!/bin/bash
timeout 900 python tools/flag_ab.py
