"""The four my_model nets and their train-step driver on the MI355X backend
(reference: web_app/components/my_model/{model,trainer,train}.py)."""
